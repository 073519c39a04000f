"""Drop-in for the reference's torch_utils/ops/fma.py: `fma(a, b, c) = a * b + c` with broadcasting.

Reference: g_nerf/torch_utils/ops/fma.py:17-60; called by the un-fused modulated convolution (training mode) as
`fma(x, dcoefs[N,C,1,1], noise)` (networks_stylegan2.py:81).  Forward = torch.addcmul, as upstream, except for that very call
shape on a GPU outside autograd, which is one pass of this repo's epilogue kernel (gnerf_modconv_epilogue: x * scale[n,c] + noise
rounded to x's dtype -- the same value torch.addcmul produces -- in x's memory format).  The gradients are the upstream ones:
da = dout * b, db = dout * a, dc = dout, each summed back to its operand's shape.
"""

import torch


def fma(a, b, c):  # => a * b + c
    return _FusedMultiplyAdd.apply(a, b, c)


def _epilogue_form(a, b, c):
    """The modulated convolution's demodulate-and-add-noise call, where the epilogue kernel applies; None otherwise."""
    if not (a.is_cuda and a.ndim == 4 and a.dtype in (torch.float16, torch.float32) and b.dtype == a.dtype and c.dtype == a.dtype):
        return None
    n, ch, h, w = a.shape
    if tuple(b.shape) != (n, ch, 1, 1) or c.ndim not in (2, 4) or tuple(c.shape[-2:]) != (h, w) or c.numel() not in (h * w, n * h * w):
        return None
    if c.ndim == 4 and (c.shape[1] != 1 or c.shape[0] not in (1, n)):
        return None
    if not (a.is_contiguous() or a.is_contiguous(memory_format=torch.channels_last)):
        return None
    import gnerf_hip
    return gnerf_hip.modconv_epilogue(a, None, scale=b.reshape(n, ch), noise=c, round_noise=False, act='linear', gain=1.0, clamp=None)


class _FusedMultiplyAdd(torch.autograd.Function):  # a * b + c
    @staticmethod
    def forward(ctx, a, b, c):
        out = None
        if not any(ctx.needs_input_grad):
            out = _epilogue_form(a, b, c)
        if out is None:
            out = torch.addcmul(c, a, b)
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        da = _unbroadcast(dout * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _unbroadcast(dout * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _unbroadcast(dout, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc


def _unbroadcast(x, shape):
    """Sum a gradient back over the axes along which its operand (of `shape`) was broadcast."""
    return x.sum_to_size(*shape) if tuple(x.shape) != tuple(shape) else x
