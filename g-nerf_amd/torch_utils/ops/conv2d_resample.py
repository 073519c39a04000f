"""Drop-in for the reference's torch_utils/ops/conv2d_resample.py: 2-D convolution with optional up/downsampling.

Reference: g_nerf/torch_utils/ops/conv2d_resample.py:48-143 (same signature, same decomposition into convolution +
`upfirdn2d`, hence the same values); called from the reference's own `modulated_conv2d` / `Conv2dLayer`
(networks_stylegan2.py:79,94,185) -- pickled or checked-out source that this repo does not replace.  What this module adds is
WHERE the pieces run on a GPU:

* fp16 activations are convolved channels_last.  MIOpen's fp16 implicit-GEMM kernels compute in NHWC and transpose in and out
  around every call when handed NCHW (profiles/r02_sr_conv_layout.jsonl: 60 -> 27, 178 -> 162, 253 -> 198, 211 -> 168 us per
  superresolution convolution at batch 1).  The result stays channels_last: this repo's `upfirdn2d` (channels_last 4x4 blur) and
  `bias_act` keep the format, so a whole fp16 SynthesisBlock runs in it after ONE layout change at its first convolution; the
  reference's block entry (`x.to(memory_format=contiguous_format)`, networks_stylegan2.py:438) changes it back.
* the 1x1 three-channel convolution of ToRGBLayer on such a tensor is one streaming read of x (gnerf_torgb_nhwc) instead of a
  MIOpen 1x1 convolution (74 -> 172 us when channels_last).
* round 6: the 3x3 convolutions themselves.  An fp16 image at a time with `groups == 1` -- what `modulated_conv2d` hands over for
  every frame of gen_videos.py, which renders ONE camera per synthesis call (gen_videos.py:154-171) -- goes to this repo's own
  implicit-GEMM kernel (csrc/conv3x3.hip: gnerf_conv3x3_epilogue_nhwc with every epilogue operand off for `up == 1`, the stride-2
  transposed form gnerf_conv_transpose3x3_s2_nhwc for `up == 2`, the blur following through the overlay's upfirdn2d as before)
  where its shape gate admits the layer (input channels in eights, output channels in blocks of 128: the four 3x3 layers of the
  256^2 / 512^2 superresolution blocks; block64's 32 -> 32 layers stay with MIOpen).  Forward only -- a tensor that feeds an
  autograd graph never comes here (`_wants_channels_last`) --, GNERF_FUSED_CONV=0 switches it off.
* CPU tensors, fp32 and anything under autograd take exactly the reference's route through torch's convolutions.

`conv2d_gradfix` stays the reference's module (resolved through the overlay's extended package path); where the reference tree is
absent (the GPU box) torch.nn.functional's convolutions are called directly, which is what conv2d_gradfix does while its
`enabled` flag is off.
"""

import os

import torch

from . import upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding
from gnerf_hip import profiled as _profiled

try:                                    # the reference's gradient-fix wrappers, when its tree is on the path
    from . import conv2d_gradfix
except ImportError:                     # GPU box / stand-alone use: plain torch convolutions
    conv2d_gradfix = None

# GNERF_CONV_CHANNELS_LAST=0 keeps the activations' layout as it arrives (A/B runs)
_CHANNELS_LAST_FP16 = os.environ.get('GNERF_CONV_CHANNELS_LAST', '1') != '0'
# GNERF_FUSED_CONV=0: every convolution goes to the framework's (MIOpen), as before round 6
_FUSED_CONV = os.environ.get('GNERF_FUSED_CONV', '1') != '0'


def _get_weight_shape(w):
    shape = [int(sz) for sz in w.shape]
    assert len(shape) == 4
    return shape


def _wants_channels_last(x, w, groups):
    """fp16 on a GPU, no autograd graph to feed, enough channels for the 16-byte channel vectors of the channels_last kernels -- and
    an ordinary convolution: the fused modulated convolution of a batch arrives as ONE image of N x C channels with `groups` = N
    (networks_stylegan2.py:91-95) and is reshaped back to [N, C_out, H, W] right after, which a channels_last result can only do
    by copying itself back to NCHW.  So batches keep the reference's layout; one image at a time (gen_videos.py's frames) and the
    un-fused form run channels_last."""
    return (_CHANNELS_LAST_FP16 and groups == 1 and x.is_cuda and x.dtype == torch.float16 and x.shape[1] % 8 == 0
            and not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)))


def _is_channels_last(x):
    return x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d / conv_transpose2d.  torch's conv2d is a correlation (flip_weight=True); flip_weight=False flips the taps."""
    _out_channels, _in_channels_per_group, kh, kw = _get_weight_shape(w)
    if not flip_weight and (kw > 1 or kh > 1):
        w = w.flip([2, 3])
    if _wants_channels_last(x, w, groups):
        if (not transpose and kh == 1 and kw == 1 and stride == 1 and groups == 1 and _out_channels == 3 and x.shape[0] == 1
                and _is_channels_last(x) and padding in (0, [0, 0], (0, 0))):
            import gnerf_hip
            if x.shape[1] in gnerf_hip.TORGB_CHANNELS:          # ToRGBLayer's convolution (w = weight * styles, networks_stylegan2.py:363)
                ones = _ones(x.shape[1], x.device)
                return gnerf_hip.torgb_channels_last(x, w.reshape(3, -1), ones)
        x = x.contiguous(memory_format=torch.channels_last)
        if _FUSED_CONV and kh == 3 and kw == 3:
            y = _own_conv3x3(x, w, stride, padding, transpose)
            if y is not None:
                return y
        w = w.contiguous(memory_format=torch.channels_last)
    if conv2d_gradfix is not None:
        op = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    else:
        op = torch.nn.functional.conv_transpose2d if transpose else torch.nn.functional.conv2d
    return op(x, w, stride=stride, padding=padding, groups=groups)


def _own_conv3x3(x, w, stride, padding, transpose):
    """x: channels_last fp16 [N, C, H, W] outside autograd, w: the weight in the form torch's operator would have taken ([O, I, 3, 3]
    correlation taps for conv2d; [I, O, 3, 3] for conv_transpose2d).  The result of csrc/conv3x3.hip's kernel, or None where its shape
    gate does not admit the call (the caller then goes to the framework as before)."""
    import gnerf_hip
    pad = [int(p) for p in padding] if isinstance(padding, (list, tuple)) else [int(padding)] * 2
    if not transpose and stride == 1 and pad == [1, 1] and gnerf_hip.conv3x3_epilogue_supported(x, w.shape[0]):
        # the bare convolution: lrelu with slope 1 and gain 1 is the identity, every other epilogue operand is absent
        return gnerf_hip.conv3x3_epilogue(x, gnerf_hip.pack_conv3x3_weights(w), alpha=1.0, gain=1.0)
    if transpose and stride == 2 and pad == [0, 0] and gnerf_hip.conv_transpose3x3_s2_supported(x, w.shape[1]):
        return gnerf_hip.conv_transpose3x3_s2(x, gnerf_hip.pack_conv_transpose3x3_weights(w.transpose(0, 1)))
    return None


_ones_cache = {}


def _ones(c, device):
    key = (c, device)
    t = _ones_cache.get(key)
    if t is None:
        t = _ones_cache[key] = torch.ones([1, c], dtype=torch.float32, device=device)
    return t


@_profiled('conv2d_resample')                       # the reference's range name (conv2d_resample.py:47, misc.profiled_function)
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """2-D convolution of x [N, C_in, H, W] with w [C_out, C_in // groups, kh, kw], optionally preceded by `up`-fold upsampling and
    followed by `down`-fold downsampling, both through the low-pass filter f (from upfirdn2d.setup_filter(); None = identity).
    `padding` (int, [x, y] or [x0, x1, y0, y1]) refers to the upsampled image and is applied once, at the beginning.
    flip_weight=False convolves instead of correlating; flip_filter likewise for f.  Returns [N, C_out, H_out, W_out]."""
    assert isinstance(x, torch.Tensor) and (x.ndim == 4)
    assert isinstance(w, torch.Tensor) and (w.ndim == 4) and (w.dtype == x.dtype)
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and (up >= 1)
    assert isinstance(down, int) and (down >= 1)
    assert isinstance(groups, int) and (groups >= 1)
    out_channels, in_channels_per_group, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # the filter's own footprint: what keeps an up- or downsampled image centred (conv2d_resample.py:82-91)
    if up > 1:
        px0, px1, py0, py1 = px0 + (fw + up - 1) // 2, px1 + (fw - up) // 2, py0 + (fh + up - 1) // 2, py1 + (fh - up) // 2
    if down > 1:
        px0, px1, py0, py1 = px0 + (fw - down + 1) // 2, px1 + (fw - down) // 2, py0 + (fh - down + 1) // 2, py1 + (fh - down) // 2

    pointwise = kw == 1 and kh == 1
    if pointwise and down > 1 and up == 1:                  # 1x1 + downsampling: resample the (smaller-channel-count-agnostic) input first
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if pointwise and up > 1 and down == 1:                  # 1x1 + upsampling: convolve at the low resolution
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                                # downsampling: low-pass, then a strided convolution
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)
    if up > 1:                                              # upsampling: stride-`up` transposed convolution, then the low-pass
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, out_channels // groups, in_channels_per_group, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * in_channels_per_group, out_channels // groups, kh, kw)
        px0, px1, py0, py1 = px0 - (kw - 1), px1 - (kw - up), py0 - (kh - 1), py1 - (kh - up)
        pxt, pyt = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x
    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:      # plain convolution with its own padding
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)

    # anything else: explicit upsampling, convolution, explicit downsampling
    x = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x
