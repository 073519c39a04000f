"""2-D resampling (upsample -> pad -> FIR -> downsample) with the reference's interface
(torch_utils/ops/upfirdn2d.py: setup_filter :72, upfirdn2d :120, _upfirdn2d_ref :168,
_upfirdn2d_cuda :219, filter2d :279, upsample2d :315, downsample2d :354; the underscore helpers
_parse_scaling / _parse_padding / _get_filter_size are imported by conv2d_resample.py:18-19).

GPU tensors run the hand-written gfx950 kernels (csrc/upfirdn2d.hip); the gradient is the same op with
up and down swapped and the filter flipped, so it runs on the same kernels.  CPU tensors, or
impl='ref', use PyTorch ops like the reference."""

import collections
import os

import numpy as np
import torch

from .. import custom_ops
from . import _fir_args

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(
            module_name='upfirdn2d_plugin',
            sources=['upfirdn2d.hip'],
            headers=['common.h'],
            source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'),
        )
    return True


# private names kept because conv2d_resample.py imports them from here (conv2d_resample.py:18-19)
_parse_scaling = _fir_args.parse_scaling
_parse_padding = _fir_args.parse_padding
_get_filter_size = _fir_args.filter_size


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Build a float32 FIR filter for upfirdn2d().

    f: list / array / tensor of shape [taps] (expanded to its outer product unless it has >= 8 taps or
    `separable` is set), [fh, fw], [] (impulse) or None (identity).  normalize makes the taps sum to 1;
    gain scales the signal (applied as gain**(ndim/2) so a separable filter gets sqrt(gain) per pass)."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2]
    assert f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f /= f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Per channel: insert up-1 zeros after each pixel, pad (negative = crop), filter with f (a true
    convolution unless flip_filter), keep every down-th pixel.

    x: [N,C,H,W] float16/32/64; f: [fh,fw], [taps] (separable) or None; up, down: int or [x, y];
    padding: int, [x, y] or [x0, x1, y0, y1], relative to the upsampled image.
    Output size per axis: (in*up + pad0 + pad1 - taps + down) // down.  Differentiable to any order."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _upfirdn2d_cuda(up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain).apply(x, f)
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)


def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """PyTorch-op implementation: explicit zero insertion, F.pad, grouped conv2d, strided slice."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    assert f.dtype == torch.float32 and not f.requires_grad
    N, C, H, W = x.shape
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    assert W * upx + padx0 + padx1 >= f.shape[-1] and H * upy + pady0 + pady1 >= f.shape[0]
    with torch.autograd.profiler.record_function('_upfirdn2d_ref'):
        x = x.reshape([N, C, H, 1, W, 1])
        x = torch.nn.functional.pad(x, [0, upx - 1, 0, 0, 0, upy - 1])
        x = x.reshape([N, C, H * upy, W * upx])
        x = torch.nn.functional.pad(x, [max(padx0, 0), max(padx1, 0), max(pady0, 0), max(pady1, 0)])
        x = x[:, :, max(-pady0, 0): x.shape[2] - max(-pady1, 0), max(-padx0, 0): x.shape[3] - max(-padx1, 0)]
        k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
        if not flip_filter:
            k = k.flip(list(range(k.ndim)))
        k = k[np.newaxis, np.newaxis].repeat([C, 1] + [1] * k.ndim)
        if k.ndim == 4:
            x = torch.nn.functional.conv2d(x, k, groups=C)
        else:
            x = torch.nn.functional.conv2d(x, k.unsqueeze(2), groups=C)
            x = torch.nn.functional.conv2d(x, k.unsqueeze(3), groups=C)
        x = x[:, :, ::downy, ::downx]
    return x


# ---------------------------------------------------------------------------------------------
# GPU path.  One autograd.Function takes the static configuration as an argument; its backward is the same
# function under the TRANSPOSED configuration (up <-> down, filter flipped, padding chosen so that the result
# has the input's shape), so gradients of any order run on the same kernels.

_Config = collections.namedtuple('_Config', 'upx upy downx downy padx0 padx1 pady0 pady1 flip gain')


def _run(x, f, c):
    """Forward computation on the plugin: one 2-D pass, or a row pass then a column pass for a separable filter."""
    if f.ndim == 2:
        return _plugin.upfirdn2d(x, f, c.upx, c.upy, c.downx, c.downy, c.padx0, c.padx1, c.pady0, c.pady1, c.flip, c.gain)
    y = _plugin.upfirdn2d(x, f.unsqueeze(0), c.upx, 1, c.downx, 1, c.padx0, c.padx1, 0, 0, c.flip, 1.0)
    return _plugin.upfirdn2d(y, f.unsqueeze(1), 1, c.upy, 1, c.downy, 0, 0, c.pady0, c.pady1, c.flip, c.gain)


def _transposed(c, f, in_hw, out_hw):
    fw, fh = _get_filter_size(f)
    (ih, iw), (oh, ow) = in_hw, out_hw
    return _Config(c.downx, c.downy, c.upx, c.upy,
                   fw - c.padx0 - 1, iw * c.upx - ow * c.downx + c.padx0 - c.upx + 1,
                   fh - c.pady0 - 1, ih * c.upy - oh * c.downy + c.pady0 - c.upy + 1,
                   not c.flip, c.gain)


def _normalise_filter(x, f):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    if f.ndim == 1 and f.shape[0] == 1:
        f = f.square().unsqueeze(0)             # a single separable tap is a 1x1 filter
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return f


def _forward_only(x, f, cfg):
    return _run(x, _normalise_filter(x, f), cfg)


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, cfg):
        f = _normalise_filter(x, f)
        y = _run(x, f, cfg)
        ctx.save_for_backward(f)
        ctx.cfg_t = _transposed(cfg, f, x.shape[2:], y.shape[2:])
        return y

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        assert not ctx.needs_input_grad[1]
        dx = _Resample.apply(dy, f, ctx.cfg_t) if ctx.needs_input_grad[0] else None
        return dx, None, None


class _Bound:
    """What `_upfirdn2d_cuda(...)` returns: `_Resample` bound to one static configuration (`.apply(x, f)`)."""

    def __init__(self, cfg):
        self.cfg = cfg

    def apply(self, x, f):
        # inference fast path: no graph wanted, so skip the autograd.Function machinery
        if not (torch.is_grad_enabled() and x.requires_grad):
            return _forward_only(x, f, self.cfg)
        return _Resample.apply(x, f, self.cfg)


_upfirdn2d_cuda_cache = dict()


def _upfirdn2d_cuda(up=1, down=1, padding=0, flip_filter=False, gain=1):
    cfg = _Config(*_parse_scaling(up), *_parse_scaling(down), *_parse_padding(padding), flip_filter, gain)
    if cfg not in _upfirdn2d_cuda_cache:
        _upfirdn2d_cuda_cache[cfg] = _Bound(cfg)
    return _upfirdn2d_cuda_cache[cfg]


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Filter without resampling; by default the output has the input's size (zeros outside the image).
    `padding` is applied on top (negative crops)."""
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsample by `up`; by default the output is exactly `up` times the input size.  The gain is
    multiplied by upx*upy to keep the signal magnitude."""
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


# ---------------------------------------------------------------------------------------------
# Producing the tri-planes in the renderer's layout.  NOT part of the reference's interface.

# Output shapes (C, H, W) for which upsample2d(up=2, 4x4 filter, float32 NCHW input on a GPU) returns a channels_last tensor:
# same values, same shape, different strides.  ImportanceRenderer fills it in with the shape of the planes it is handed when
# GNERF_NHWC_PLANES=1 (opt-in: in the reference's SynthesisBlock the next op is `img.add_(y)` with an NCHW `y`, a strided
# add), so that the backbone's last skip-image upsample writes the planes' memory as [N, H, W, 96] and the renderer reads it
# without a layout change.  This repo's own generator does not need the hint: it calls upsample2d_add_channels_last.
channels_last_output_shapes = set()


class _UpsampleAddChannelsLast(torch.autograd.Function):
    """upsample2d(img, f) + y in one launch, written channels_last (gnerf_hip.upsample2x_add_nhwc)."""

    @staticmethod
    def forward(ctx, img, y, f, flip_filter, gain):
        import gnerf_hip
        out, amax = gnerf_hip.upsample2x_add_nhwc(img, y, f, flip=flip_filter, gain=gain, with_absmax=True)
        # max |out|, for the renderer's decoder-arithmetic choice.  Inference tensors (torch.inference_mode) have no version
        # counter to validate the tag against: they go untagged and the render launcher measures max |planes| itself.
        if not out.is_inference():
            out._gnerf_absmax = (out._version, amax)
        ctx.save_for_backward(f)
        ctx.has_y = y is not None
        ctx.cfg_t = _transposed(_Config(2, 2, 1, 1, 2, 1, 2, 1, flip_filter, gain), f, img.shape[2:], out.shape[2:])
        return out

    @staticmethod
    def backward(ctx, d_out):
        (f,) = ctx.saved_tensors
        d_img = d_y = None
        if ctx.needs_input_grad[0]:             # the adjoint of the x2 upsample: the transposed configuration, as in _Resample
            d_img = _Resample.apply(d_out.contiguous(), f, ctx.cfg_t)
        if ctx.has_y and ctx.needs_input_grad[1]:
            d_y = d_out
        return d_img, d_y, None, None, None


def upsample2d_add_channels_last(img, y, f, flip_filter=False, gain=1):
    """upsample2d(img, f, up=2, gain=gain) + y (y may be None), as a [N,C,2H,2W] tensor whose MEMORY is channels_last
    ([N,2H,2W,C]) -- for C = 96 exactly the interleaved tri-plane layout the fused renderer reads in place.  One kernel on a
    GPU for float32 NCHW inputs with C % 32 == 0; anything else composes upsample2d and the add (NCHW result)."""
    if img.device.type == 'cuda' and img.dtype == torch.float32 and isinstance(f, torch.Tensor) and tuple(f.shape) == (4, 4) \
            and img.is_contiguous() and img.shape[1] % 32 == 0 and img.shape[3] % 16 == 0 and img.shape[2] % 2 == 0 \
            and (y is None or (y.dtype == torch.float32 and y.is_contiguous() and y.shape[1:] == (img.shape[1], 2 * img.shape[2], 2 * img.shape[3]))) and _init():
        if torch.is_grad_enabled() and (img.requires_grad or (y is not None and y.requires_grad)):
            return _UpsampleAddChannelsLast.apply(img, y, f, flip_filter, gain * 4)
        return _UpsampleAddChannelsLast.forward(_NoCtx(), img, y, f, flip_filter, gain * 4)
    out = upsample2d(img, f, flip_filter=flip_filter, gain=gain)
    return out if y is None else out.add_(y) if not (torch.is_grad_enabled() and (out.requires_grad or y.requires_grad)) else out + y


class _NoCtx:
    def save_for_backward(self, *a):
        pass


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsample by `down`; by default the output is exactly 1/down of the input size."""
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2, pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
