"""Filtered leaky ReLU (bias -> upsample FIR -> lrelu/clamp -> downsample FIR) with the reference's
interface (torch_utils/ops/filtered_lrelu.py: filtered_lrelu :58, _filtered_lrelu_ref :122,
_filtered_lrelu_cuda :161).

G-NeRF never executes this op (only StyleGAN3's SynthesisLayer calls it and G-NeRF never builds one);
it is kept API-complete.  On a GPU it runs as ONE hand-written gfx950 launch (csrc/filtered_lrelu_fused.hip:
separable filters, up/down in {1,2,4}, <= 8 taps per polyphase branch, fp16/fp32) with the upsampled
intermediate held in LDS; configurations the fused kernel declines (return code -1) run as three launches
-- upfirdn2d, filtered_lrelu_act_ (which also records the 2-bit sign tensor), upfirdn2d -- the path the
reference itself takes whenever its fused kernel has no specialisation (filtered_lrelu.py:225-231).
Only the packed sign tensor is kept for the backward pass, which is the same op with up and down
swapped reading the signs back."""

import collections
import os
import warnings

import numpy as np
import torch

from .. import custom_ops
from . import upfirdn2d
from . import _fir_args
from . import bias_act

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(
            module_name='filtered_lrelu_plugin',
            sources=['filtered_lrelu.hip'],
            headers=['common.h'],
            source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'),
        )
    return True


def _get_filter_size(f):
    return _fir_args.filter_size(f, strict=False)      # width, height


def _parse_padding(padding):
    return _fir_args.parse_padding(padding, kinds=_fir_args.NUMPY_INTS)


def _check_scalars(up, down, gain, slope, clamp):
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False, impl='cuda'):
    """x [N,C,H,W] -> add bias b, zero-upsample by `up`, pad, filter with fu (gain up**2), multiply by
    `gain`, leaky ReLU with `slope`, clamp to +-clamp, filter with fd, keep every `down`-th pixel.
    Output size per axis: (in*up + pad0 + pad1 - (fu_taps-1) - (fd_taps-1) + (down-1)) // down."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _filtered_lrelu_cuda(up=up, down=down, padding=padding, gain=gain, slope=slope, clamp=clamp,
                                    flip_filter=flip_filter).apply(x, fu, fd, b, None, 0, 0)
    return _filtered_lrelu_ref(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=padding, gain=gain, slope=slope, clamp=clamp, flip_filter=flip_filter)


def _filtered_lrelu_ref(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    """Composition of bias_act() and upfirdn2d()."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype
        assert list(b.shape) == [x.shape[1]]
    _check_scalars(up, down, gain, slope, clamp)
    px0, px1, py0, py1 = _parse_padding(padding)
    N, C, in_h, in_w = x.shape
    in_dtype = x.dtype
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    with torch.autograd.profiler.record_function('_filtered_lrelu_ref'):
        x = bias_act.bias_act(x=x, b=b)
        x = upfirdn2d.upfirdn2d(x=x, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
        x = bias_act.bias_act(x=x, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
        x = upfirdn2d.upfirdn2d(x=x, f=fd, down=down, flip_filter=flip_filter)
    assert list(x.shape) == [N, C, out_h, out_w]
    assert x.dtype == in_dtype
    return x


# ---------------------------------------------------------------------------------------------
# GPU path.  As in upfirdn2d.py, one Function takes its static configuration as an argument and its backward is the
# same Function under the transposed configuration; what links forward and backward is the packed sign tensor.

_Config = collections.namedtuple('_Config', 'up down px0 px1 py0 py1 gain slope clamp flip')


class _FilteredLRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fu, fd, b, si, sx, sy, c):
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        dev = x.device
        fu = torch.ones([1, 1], dtype=torch.float32, device=dev) if fu is None else fu
        fd = torch.ones([1, 1], dtype=torch.float32, device=dev) if fd is None else fd
        assert 1 <= fu.ndim <= 2 and 1 <= fd.ndim <= 2
        if c.up == 1 and fu.ndim == 1 and fu.shape[0] == 1:
            fu = fu.square()[None]
        if c.down == 1 and fd.ndim == 1 and fd.shape[0] == 1:
            fd = fd.square()[None]
        si = torch.empty([0]) if si is None else si
        b = torch.zeros([x.shape[1]], dtype=x.dtype, device=dev) if b is None else b
        write_signs = si.numel() == 0 and (x.requires_grad or b.requires_grad)
        strides = [x.stride(i) for i in range(x.ndim) if x.size(i) > 1]
        if any(lo < hi for lo, hi in zip(strides[:-1], strides[1:])):
            warnings.warn("low-performance memory layout detected in filtered_lrelu input", RuntimeWarning)
        pad = [c.px0, c.px1, c.py0, c.py1]
        # The fused single-launch kernel first; return code < 0 = it does not cover this configuration, and the
        # three-launch route below (the reference's route for configurations ITS fused kernel lacks) takes over.
        y, so, rc = _plugin.filtered_lrelu(x, fu, fd, b, si, c.up, c.down, *pad, sx, sy, c.gain, c.slope, c.clamp, c.flip, write_signs)
        if rc < 0:
            y = x.add(b.unsqueeze(-1).unsqueeze(-1))
            y = upfirdn2d.upfirdn2d(x=y, f=fu, up=c.up, padding=pad, gain=c.up ** 2, flip_filter=c.flip)
            so = _plugin.filtered_lrelu_act_(y, si, sx, sy, c.gain, c.slope, c.clamp, write_signs)      # in place on y
            y = upfirdn2d.upfirdn2d(x=y, f=fd, down=c.down, flip_filter=c.flip)
        ctx.save_for_backward(fu, fd, si if si.numel() else so)
        ctx.cfg, ctx.in_hw, ctx.out_hw, ctx.sign_offset = c, tuple(x.shape[2:]), tuple(y.shape[2:]), (sx, sy)
        return y

    @staticmethod
    def backward(ctx, dy):
        fu, fd, si = ctx.saved_tensors
        c, (xh, xw), (yh, yw), (sx, sy) = ctx.cfg, ctx.in_hw, ctx.out_hw, ctx.sign_offset
        assert not any(ctx.needs_input_grad[i] for i in (1, 2, 4, 5, 6))
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
            fu_w, fu_h, fd_w, fd_h = fu.shape[-1], fu.shape[0], fd.shape[-1], fd.shape[0]
            t = _Config(up=c.down, down=c.up,
                        px0=(fu_w - 1) + (fd_w - 1) - c.px0, px1=xw * c.up - yw * c.down + c.px0 - (c.up - 1),
                        py0=(fu_h - 1) + (fd_h - 1) - c.py0, py1=xh * c.up - yh * c.down + c.py0 - (c.up - 1),
                        gain=c.gain * (c.up ** 2) / (c.down ** 2), slope=c.slope, clamp=float('inf'), flip=not c.flip)
            dx = _FilteredLRelu.apply(dy, fd, fu, None, si, sx - (fu_w - 1) + c.px0, sy - (fu_h - 1) + c.py0, t)
        if ctx.needs_input_grad[3]:
            db = dx.sum([0, 2, 3])
        return dx, None, None, db, None, None, None, None


class _Bound:
    """What `_filtered_lrelu_cuda(...)` returns: the Function bound to one configuration (`.apply(x, fu, fd, b, si, sx, sy)`)."""

    def __init__(self, cfg):
        self.cfg = cfg

    def apply(self, x, fu, fd, b, si, sx, sy):
        return _FilteredLRelu.apply(x, fu, fd, b, si, sx, sy, self.cfg)


_filtered_lrelu_cuda_cache = dict()


def _filtered_lrelu_cuda(up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    _check_scalars(up, down, gain, slope, clamp)
    cfg = _Config(up, down, *_parse_padding(padding), float(gain), float(slope), float('inf' if clamp is None else clamp), flip_filter)
    if cfg not in _filtered_lrelu_cuda_cache:
        _filtered_lrelu_cuda_cache[cfg] = _Bound(cfg)
    return _filtered_lrelu_cuda_cache[cfg]
