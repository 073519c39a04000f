"""Tri-plane generator and depth discriminator for benchmarks, tests and harnesses that must run where the reference tree is absent
(the GPU box): the CALLERS on either side of the hot path -- StyleGAN2 backbone -> tri-planes -> [renderer] ->
super-resolution to 512x512 -- as plain PyTorch modules (convolutions go to MIOpen) around this repo's renderer and custom
ops.  SURVEY.md section 8a rows 11 and C: these stay PyTorch; nothing here is a kernel.

The layer graph and the parameter / buffer NAMES are those of the reference's FFHQ configuration, so a reference
state_dict loads with strict=True and the two produce the same images (tests/test_generator_cpu.py does exactly that in
the build container):
    TriPlaneGenerator         g_nerf/training/triplane.py:19-108          -> Generator
    OSGDecoder                g_nerf/training/triplane.py:113-136         -> gnerf_harness.TriPlaneDecoder
    Generator/Mapping/Synthesis*  g_nerf/training/networks_stylegan2.py:41-557  -> Backbone, Mapping, Synthesis, Block, StyledConv, ToRGB
    conv2d_resample (up=2)    g_nerf/torch_utils/ops/conv2d_resample.py:114-131 -> StyledConv.forward
    SuperresolutionHybrid8XDC g_nerf/training/superresolution.py:266-303  -> SuperRes8XDC
    Discriminator (+ blocks, epilogue, minibatch-std)  g_nerf/training/networks_stylegan2.py:561-799 -> Discriminator, DBlock, DEpilogue
Both execution modes of the reference's layers exist: the fused form (per-sample modulated weights in one grouped
convolution: inference, `fused_modconv=True`) and the un-fused form that training uses under the FFHQ configuration's
`fused_modconv_default='inference_only'` (train.py:304, networks_stylegan2.py:76-86: activations scaled by the styles,
ONE shared-weight convolution, demodulation and noise applied after it), with 'random' / 'const' / 'none' noise.
What is deliberately missing: EMA updates and truncation of the mapping network, the 'orig' block architecture, the other
seven super-resolution variants, Freeze-D, pickling hooks.
"""

import math
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from torch_utils.ops import bias_act, conv2d_resample, fma, upfirdn2d
from training.volumetric_rendering.renderer import ImportanceRenderer
from training.volumetric_rendering.ray_sampler import RaySampler

import gnerf_harness as H

LRELU_GAIN = math.sqrt(2)
# Inference on a GPU: the ATen elementwise chains around each convolution (weight modulation / demodulation, noise, bias,
# activation, clamp: ~10 launches per layer) run as this repo's modconv kernels (csrc/modconv.hip), and a
# batch of fp16 layers uses the shared-weight form of the convolution (one batched MIOpen convolution instead of a grouped one
# with per-sample weights: 2.2x faster at batch 4, tools/bench_upconv.py).  GNERF_MODCONV_FAST=0 keeps the plain PyTorch forms.
_MODCONV_FAST = os.environ.get('GNERF_MODCONV_FAST', '1') != '0'


# The fp16 blocks run channels_last on the fast path (memory [N,H,W,C]): MIOpen's fp16 convolutions compute in that layout and
# otherwise transpose in and out around every call (tools/bench_sr_conv_layout.py: 0.63 ms of the superresolution at batch 4), and
# the surrounding kernels (blur, epilogue, ToRGB) have channels_last forms.  GNERF_FP16_CHANNELS_LAST=0 keeps NCHW.  Inference fast path
# only: under autograd (un-fused forms) the same switch measured G forward 12.7 -> 13.4 ms, G backward 26.6 -> 25.8 ms: no gain.
_FP16_CHANNELS_LAST = os.environ.get('GNERF_FP16_CHANNELS_LAST', '1') != '0'


def _is_channels_last(x):
    return x.ndim == 4 and x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def _fast_path(x, *params):
    return (_MODCONV_FAST and x.is_cuda and (x.is_contiguous() or _is_channels_last(x)) and x.dtype in (torch.float16, torch.float32)
            and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))))


def _cast_param(module, name, dtype):
    """getattr(module, name) in `dtype`, the converted copy cached per parameter version: the fp16 layers' biases are fp32 parameters,
    and converting them inside every kernel wrapper was ten 4-us copy launches per orbit frame (4 % of its GPU time,
    profiles/r03_orbit_fast_flow_kernel_stats.csv)."""
    p = getattr(module, name)
    if p is None or p.dtype == dtype:
        return p
    if p.is_inference() or (torch.is_grad_enabled() and p.requires_grad):
        return p.to(dtype)
    key = (p.data_ptr(), p._version, dtype)
    cache = module.__dict__.setdefault('_gnerf_cast', {})
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = (key, p.detach().to(dtype).contiguous())
        cache[name] = hit
    return hit[1]


def _prenormalised_weight(module, dtype, channels_last=False, transposed=False):
    """weight / (sqrt(fan_in) max|weight[o]|) in `dtype` (networks_stylegan2.py:63), cached per weight version: a constant at inference.
    channels_last: in that memory format (for channels_last activations); transposed: as [I,O,k,k] for conv_transpose2d."""
    w = module.weight
    key = (w.data_ptr(), w._version if not w.is_inference() else None, dtype, channels_last, transposed)
    cache = module.__dict__.setdefault('_gnerf_prenorm', {})
    hit = cache.get(key[2:])
    if hit is None or hit[0] != key:
        with torch.no_grad():
            v = (w * (1 / math.sqrt(w[0].numel()) / w.norm(float('inf'), dim=[1, 2, 3], keepdim=True))).to(dtype)
            if transposed:
                v = v.transpose(0, 1)
            hit = (key, v.contiguous(memory_format=torch.channels_last if channels_last else torch.contiguous_format))
        cache[key[2:]] = hit
    return hit[1]


def _packed_prenormalised_weight(module, dtype, transposed=False):
    """The pre-normalised weight in the tap-major [9, O, I] form of gnerf_hip.conv3x3_epilogue (csrc/conv3x3.hip) -- transposed: in the
    by-output-phase form of gnerf_hip.conv_transpose3x3_s2 -- cached like the convolution forms above."""
    w = module.weight
    key = (w.data_ptr(), w._version if not w.is_inference() else None, dtype)
    name = '_gnerf_prenorm_packed_t' if transposed else '_gnerf_prenorm_packed'
    hit = module.__dict__.get(name)
    if hit is None or hit[0] != key:
        import gnerf_hip
        pack = gnerf_hip.pack_conv_transpose3x3_weights if transposed else gnerf_hip.pack_conv3x3_weights
        hit = (key, pack(_prenormalised_weight(module, dtype), dtype))
        module.__dict__[name] = hit
    return hit[1]


def _packed_modulated_weight(module, nstyles, dtype, transposed=False):
    """pack(f16(pre-normalised weight x the ONE latent's normalised styles)): the input scaling of a layer folded into its packed weights, which is
    the order the reference's own inference path takes it in (the fused form modulates the WEIGHTS, networks_stylegan2.py:66-75) -- one rounding
    per weight instead of one per activation, and no scaling pass over the activations.  A constant of (latent, weight): cached by _per_latent."""
    import gnerf_hip
    w = module.weight
    with torch.no_grad():
        v = w * (1 / math.sqrt(w[0].numel()) / w.norm(float('inf'), dim=[1, 2, 3], keepdim=True)) * nstyles.reshape(1, -1, 1, 1).to(w.dtype)
    pack = gnerf_hip.pack_conv_transpose3x3_weights if transposed else gnerf_hip.pack_conv3x3_weights
    return pack(v.to(dtype), dtype)


# GNERF_LATENT_WEIGHTS=0: a layer whose input scaling no earlier epilogue carries (the first layer of a block) scales its activations in a launch
# of its own (gnerf_scale_channels), as until round 6, also when the whole batch shares ONE latent (an orbit: k views of one object)
_LATENT_WEIGHTS = os.environ.get('GNERF_LATENT_WEIGHTS', '1') != '0'
# GNERF_FUSED_TORGB=0: the last block of the superresolution keeps its three launches (layer, ToRGB added to the running image) instead of the
# layer's convolution with the ToRGB in its epilogue and no layer output at all (gnerf_conv3x3_epilogue_torgb_nhwc, round 6)
_FUSED_TORGB = os.environ.get('GNERF_FUSED_TORGB', '1') != '0'
_TORGB_DONE = object()             # what StyledConv.forward(..., torgb_tail=...) returns when the tail ran in the convolution's launch
# GNERF_FUSED_CONV=0: the 3x3 layers of the shared-weight form go to MIOpen + gnerf_modconv_epilogue_nhwc (round 4's flow) instead of
# the one-launch kernel of csrc/conv3x3.hip
_FUSED_CONV = os.environ.get('GNERF_FUSED_CONV', '1') != '0'
# GNERF_SHARED_AT_ONE=0: a batch of one keeps round 5's route (per-sample modulated weights into the framework's convolution)
_SHARED_AT_ONE = os.environ.get('GNERF_SHARED_AT_ONE', '1') != '0'
# GNERF_F32X3=0: the backbone's float32 3x3 layers stay with the framework's fp32 convolution (MIOpen) instead of the fp32-grade form of
# csrc/conv3x3.hip (round 6: every product as three f16 matrix products of hi / lo splits, 2.5-3.3x MIOpen on the backbone's hot shapes)
_F32X3 = os.environ.get('GNERF_F32X3', '1') != '0'
# smallest image the fp32-grade form takes: the convolution at H, W >= 64, the x2 layers from H, W >= 32 (at 32^2 the kernel's 8 x 32 pixel
# tiles leave most of the chip idle and MIOpen is faster: profiles/r06_conv_f32grade_gate.jsonl)
_F32X3_MIN_CONV, _F32X3_MIN_UP = 64, 32


def _packed_weight_f32x3(module, transposed=False):
    """The RAW float32 weight (no pre-normalisation: float32 layers have none, networks_stylegan2.py:61-64) split as [hi | hi | lo] and packed for
    gnerf_hip.conv3x3_f32x3_epilogue (or, transposed, conv_transpose3x3_s2_f32x3); cached per weight version."""
    w = module.weight
    key = (w.data_ptr(), w._version if not w.is_inference() else None)
    name = '_gnerf_f32x3_packed_t' if transposed else '_gnerf_f32x3_packed'
    hit = module.__dict__.get(name)
    if hit is None or hit[0] != key:
        import gnerf_hip
        pack = gnerf_hip.pack_conv_transpose3x3_weights_f32x3 if transposed else gnerf_hip.pack_conv3x3_weights_f32x3
        hit = (key, pack(w))
        module.__dict__[name] = hit
    return hit[1]


def _latent_token(w):
    """What identifies a latent slice `w` (a view of the caller's ws) for caches of values that depend on it and on parameters alone:
    (weak reference to the tensor that owns the memory, its version counter, the view's offset and shape) -- or None when changes
    cannot be tracked (inference tensors have no version counter) or the values must stay in an autograd graph."""
    base = w._base if w._base is not None else w
    if base.is_inference() or (torch.is_grad_enabled() and base.requires_grad):
        return None
    return weakref.ref(base), (base._version, w.storage_offset(), tuple(w.shape), w.dtype)


def _per_latent(module, w, tag, params, fn):
    """fn() memoised per (latent slice, parameter versions) on `module`.  gen_videos.py renders a whole orbit from ONE ws
    (gen_videos.py:150), so every style vector, every modulated weight tensor and every demodulation coefficient of the
    superresolution layers is a constant of the orbit: without this each frame recomputes them (modulate_weights_kernel alone was
    4.6 % of the GPU time of profiles/r02_generator_kernel_stats.csv).  The cached tensors are never written again; a FrameProgram
    (gen_videos_mi355x.py) keeps references to the ones its graph captured."""
    tok = _latent_token(w)
    if tok is None or (torch.is_grad_enabled() and any(p.requires_grad for p in params)) or any(p.is_inference() for p in params):
        return fn()
    ref, state = tok
    key = (state, tuple((p.data_ptr(), p._version) for p in params))
    cache = module.__dict__.setdefault('_gnerf_latent_cache', {})
    hit = cache.get(tag)
    base = ref()
    if hit is not None and hit[0]() is base and hit[1] == key:
        return hit[2]
    with torch.no_grad():
        val = fn()
    cache[tag] = (ref, key, val)
    return val


# Latent tensors whose rows are copies of ONE latent (Generator.synthesis makes them for k views of one object; the superresolution's `ws3` inherits
# the mark): what lets a layer fold per-latent constants into its weights without reading the rows back from the device.
_ONE_LATENT = {}                   # id(tensor) -> weak reference (tensors compare element-wise: no WeakSet)


def _mark_one_latent(t):
    key = id(t)
    if key not in _ONE_LATENT or _ONE_LATENT[key]() is not t:
        _ONE_LATENT[key] = weakref.ref(t, lambda _, k=key: _ONE_LATENT.pop(k, None))


def _rows_share_latent(w):
    base = w._base if w._base is not None else w
    ref = _ONE_LATENT.get(id(base))
    return w.shape[0] == 1 or (ref is not None and ref() is base)


def _latent_cacheable(w, params):
    """True when _per_latent(module, w, tag, params, fn) would keep fn()'s value (the conditions of its first lines)."""
    return (_latent_token(w) is not None and not (torch.is_grad_enabled() and any(p.requires_grad for p in params))
            and not any(p.is_inference() for p in params))


def clear_latent_caches(root):
    """Drop every per-latent / per-parameter cache under `root` (modulated weights [N,O,I,3,3] per layer are ~38 MB per 512-channel
    fp32 layer at batch 4: a few hundred MB over the backbone, held until another latent replaces them).  Call it when an orbit is
    done and the memory is wanted back; a FrameProgram that captured the cached tensors keeps its own references and stays valid.
    The caches assume ONE stream per generator: the tensors are made on whichever stream first asked for them and later read in
    place (a HIP-graph capture of the same generator warms them up on its side stream before it captures: FrameProgram.__init__)."""
    for m in root.modules():
        for k in ('_gnerf_latent_cache', '_gnerf_prenorm', '_gnerf_cast'):
            m.__dict__.pop(k, None)


def latent_cache_tensors(root):
    """Every tensor the per-latent caches under `root` currently hold (for a FrameProgram to keep alive)."""
    out = []
    for m in root.modules():
        for _, _, val in m.__dict__.get('_gnerf_latent_cache', {}).values():
            out.extend(v for v in (val if isinstance(val, (tuple, list)) else (val,)) if isinstance(v, torch.Tensor))
    return out


class Linear(nn.Module):
    """FullyConnectedLayer (networks_stylegan2.py:101-134): runtime-scaled weight, optional activation through bias_act."""

    def __init__(self, n_in, n_out, activation='linear', lr_multiplier=1.0, bias_init=0.0):
        super().__init__()
        self.activation = activation
        self.weight = nn.Parameter(torch.randn(n_out, n_in) / lr_multiplier)
        self.bias = nn.Parameter(torch.full([n_out], float(bias_init)))
        self.weight_gain, self.bias_gain = lr_multiplier / math.sqrt(n_in), lr_multiplier

    def _scaled(self, dtype):
        """(weight * weight_gain, bias * bias_gain) in `dtype`; constants at inference, cached per parameter version."""
        w, b = self.weight, self.bias
        if torch.is_grad_enabled() and (w.requires_grad or b.requires_grad):
            return w.to(dtype) * self.weight_gain, b.to(dtype) * self.bias_gain
        key = (w.data_ptr(), b.data_ptr(), None if w.is_inference() else (w._version, b._version), dtype)
        cache = self.__dict__.get('_gnerf_scaled')
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, (w.to(dtype) * self.weight_gain).contiguous(), (b.to(dtype) * self.bias_gain).contiguous())
            self.__dict__['_gnerf_scaled'] = cache
        return cache[1], cache[2]

    def forward(self, x):
        w, b = self._scaled(x.dtype)
        if self.activation == 'linear':
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)


def _second_moment_normalize(x, eps=1e-8):
    return x * (x.square().mean(dim=1, keepdim=True) + eps).rsqrt()


class Mapping(nn.Module):
    """MappingNetwork (networks_stylegan2.py:200-272) without truncation / EMA: z, c -> ws [N, num_ws, w_dim]."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=2, lr_multiplier=0.01):
        super().__init__()
        self.num_ws, self.num_layers = num_ws, num_layers
        self.embed = Linear(c_dim, w_dim)
        feats = [z_dim + w_dim] + [w_dim] * num_layers
        for i in range(num_layers):
            setattr(self, f'fc{i}', Linear(feats[i], feats[i + 1], activation='lrelu', lr_multiplier=lr_multiplier))
        self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c):
        x = torch.cat([_second_moment_normalize(z.float()), _second_moment_normalize(self.embed(c.float()))], dim=1)
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        return x.unsqueeze(1).repeat(1, self.num_ws, 1)


def _prenormalize(weight, styles):
    """Scaling of weights and styles against fp16 overflow, cancelled by the demodulation (networks_stylegan2.py:62-64)."""
    weight = weight * (1 / math.sqrt(weight[0].numel()) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
    return weight, styles / styles.norm(float('inf'), dim=1, keepdim=True)


def _modulated_weights(weight, styles, demodulate, half):
    """Per-sample convolution weights [N, O, I, k, k] (networks_stylegan2.py:61-75, the fused form used at inference)."""
    if half and demodulate:
        weight, styles = _prenormalize(weight, styles)
    w = weight.unsqueeze(0) * styles[:, None, :, None, None]
    if demodulate:
        w = w * (w.square().sum(dim=[2, 3, 4], keepdim=True) + 1e-8).rsqrt()
    return w


def _demod_coefficients(weight, styles):
    """[N, O] factors that normalise each output channel of the modulated weights (networks_stylegan2.py:71-72), without
    materialising the [N, O, I, k, k] product: sum_ikk (w s)^2 = (w^2 summed over the taps) . s^2."""
    return (weight.square().sum(dim=[2, 3]).matmul(styles.square().t()).t() + 1e-8).rsqrt()


class StyledConv(nn.Module):
    """SynthesisLayer (networks_stylegan2.py:280-345): modulated 3x3 convolution (optionally x2 upsampling = transposed
    convolution + the 4x4 blur, conv2d_resample.py:114-131), noise, bias + leaky ReLU (+ clamp) in bias_act."""

    def __init__(self, c_in, c_out, w_dim, resolution, up=1, conv_clamp=None):
        super().__init__()
        self.up, self.conv_clamp, self.resolution = up, conv_clamp, resolution
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))
        self.affine = Linear(w_dim, c_in, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn(c_out, c_in, 3, 3))
        self.register_buffer('noise_const', torch.randn(resolution, resolution))
        self.noise_strength = nn.Parameter(torch.zeros([]))
        self.bias = nn.Parameter(torch.zeros(c_out))

    def _resampled_conv(self, x, weight, groups, weight_t=None, epilogue=None):
        """weight [groups*O, I, 3, 3] (correlation form).  up == 1: 3x3 convolution with padding 1.  up == 2: stride-2
        transposed convolution (kernel as is: the reference un-flips it twice), 2H+1 outputs per axis, then the low-pass
        filter with gain up^2 and one pixel of padding -> 2H.  weight_t: the [groups*I, O, 3, 3] form, when the caller has it.
        epilogue: keyword arguments of gnerf_hip.modconv_epilogue (no noise); the call then returns (x, done) and, where the fused
        blur + epilogue kernel applies (x2 layers on channels_last activations), x already carries the epilogue (done = True)."""
        fmt = torch.channels_last if _is_channels_last(x) else torch.contiguous_format
        if self.up == 1:
            x = F.conv2d(x, weight if fmt == torch.contiguous_format else weight.contiguous(memory_format=fmt), padding=1, groups=groups)
            return (x, False) if epilogue is not None else x
        if weight_t is not None and not torch.is_tensor(weight_t):          # ('phases', packed): the x2 layer on csrc/conv3x3.hip's transposed form
            import gnerf_hip
            x = gnerf_hip.conv_transpose3x3_s2(x, weight_t[1])
            weight_t = False
        if weight_t is None:
            o, i = weight.shape[0] // groups, weight.shape[1]
            weight_t = weight.reshape(groups, o, i, 3, 3).transpose(1, 2).reshape(groups * i, o, 3, 3)
            if fmt == torch.channels_last:
                weight_t = weight_t.contiguous(memory_format=fmt)
        if weight_t is not False:
            x = F.conv_transpose2d(x, weight_t, stride=2, groups=groups)
        if epilogue is not None and groups == 1 and _is_channels_last(x) and x.shape[1] % 8 == 0:
            import gnerf_hip                                # blur + epilogue in one pass over the activations (csrc/upfirdn2d.hip)
            return gnerf_hip.blur_epilogue_channels_last(x, self.resample_filter, [1, 1, 1, 1], blur_gain=4, **epilogue), True
        x = upfirdn2d.upfirdn2d(x, self.resample_filter, padding=[1, 1, 1, 1], gain=4)
        return (x, False) if epilogue is not None else x

    def forward(self, x, w, noise_mode='random', gain=1.0, fused=True, prescaled=False, next_layer=None, next_w=None, torgb_tail=None):
        """prescaled: x already carries this layer's input scaling (see next_layer).  next_layer / next_w: the StyledConv that
        consumes the result and its w; where the shared-weight channels_last form applies, that layer's `x * styles` is folded
        into this layer's epilogue, and the call returns (x, folded) instead of x.
        torgb_tail = (ToRGB module, its w, img): the caller needs only `img += torgb(layer(x))` (a block whose x nothing else reads); where the
        one-launch form applies (gnerf_hip.conv3x3_epilogue_torgb) it runs, img is updated in place and the call returns _TORGB_DONE --
        otherwise x as always, and the caller goes on as without the argument."""
        assert noise_mode in ('random', 'const', 'none')
        n, c_in, h, wd = x.shape
        aff = (self.affine.weight, self.affine.bias)
        fast = _fast_path(x, w, self.weight, self.bias, self.noise_strength, *aff)      # w: a latent that needs a gradient takes the autograd forms
        styles = _per_latent(self, w, 'styles', aff, lambda: self.affine(w)) if fast else self.affine(w)
        noise = None
        if noise_mode == 'random':
            noise = torch.randn([n, 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        elif noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        folded = False
        assert not prescaled or (fast and x.dtype == torch.float16 and (n > 1 or _SHARED_AT_ONE))
        if fast:
            import gnerf_hip
            half = x.dtype == torch.float16
            cl = _is_channels_last(x)
            c_out = self.weight.shape[0]
            clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
            mine = aff + (self.weight,)
            if (not half and _F32X3 and c_out % 128 == 0 and c_in % 8 == 0 and (noise is None or noise.numel() == (h * self.up) * (wd * self.up))
                    and ((self.up == 1 and min(h, wd) >= _F32X3_MIN_CONV and gnerf_hip.conv3x3_f32x3_supported(x, c_out))
                         or (self.up == 2 and min(h, wd) >= _F32X3_MIN_UP and gnerf_hip.conv_transpose3x3_s2_f32x3_supported(x, c_out)))):
                # float32 layer in fp32-GRADE arithmetic on the f16 matrix cores (csrc/conv3x3.hip, OUT32): the reference's un-fused form --
                # x * styles, convolution with the layer's weight, demodulation coefficients after it (networks_stylegan2.py:76-83) -- with the
                # scaling folded into the hi / lo split of the activations and everything behind the convolution in its epilogue
                dco = _per_latent(self, w, 'dco32', mine, lambda: gnerf_hip.modulate_weights(self.weight, styles, True, out_dtype=torch.float32, want_weights=False, want_dcoefs=True)[1])
                x3 = gnerf_hip.split_f16x3(x if cl else x.contiguous(memory_format=torch.channels_last), styles)
                if self.up == 1:
                    x = gnerf_hip.conv3x3_f32x3_epilogue(x3, _packed_weight_f32x3(self), self.bias, scale=dco, noise=noise, gain=LRELU_GAIN * gain, clamp=clamp)
                else:
                    x = gnerf_hip.conv_transpose3x3_s2_f32x3(x3, _packed_weight_f32x3(self, transposed=True))
                    if noise is None:
                        x = gnerf_hip.blur_epilogue_channels_last(x, self.resample_filter, [1, 1, 1, 1], blur_gain=4, bias=self.bias, scale=dco, act='lrelu',
                                                                  gain=LRELU_GAIN * gain, clamp=clamp)
                    else:
                        x = upfirdn2d.upfirdn2d(x, self.resample_filter, padding=[1, 1, 1, 1], gain=4)
                        x = gnerf_hip.modconv_epilogue(x, self.bias, scale=dco, noise=noise, act='lrelu', gain=LRELU_GAIN * gain, clamp=clamp)
                return (x, folded) if next_layer is not None else x
            # shared-weight form: activations scaled by the styles, demodulation in the epilogue.  Round 6: also at n == 1 -- one latent is
            # trivially "shared", and it is the call gen_videos.py makes (one camera per synthesis, gen_videos.py:154-171): the frame-by-frame
            # orbit then runs its 3x3 layers on csrc/conv3x3.hip like the batched one instead of per-sample weights + MIOpen
            if half and (n > 1 or _SHARED_AT_ONE):
                dco = _per_latent(self, w, 'dco', mine, lambda: gnerf_hip.modulate_weights(self.weight, styles, True, out_dtype=x.dtype, want_weights=False, want_dcoefs=True)[1])
                own_plain = _FUSED_CONV and self.up == 1 and cl and gnerf_hip.conv3x3_epilogue_supported(x, c_out) and (noise is None or noise.numel() == h * wd)
                own_up = self.up == 2 and _FUSED_CONV and cl and gnerf_hip.conv_transpose3x3_s2_supported(x, c_out)
                packed = None           # the layer's packed weights when they carry its input scaling (round 6: ONE latent for the whole batch)
                if not prescaled:
                    nst = _per_latent(self, w, 'nstyles', aff, lambda: gnerf_hip.normalise_styles(styles))
                    if _LATENT_WEIGHTS and _rows_share_latent(w) and (own_plain or own_up) and _latent_cacheable(w, mine):     # (packing per call would cost more than the scaling pass)
                        packed = _per_latent(self, w, ('packed_mod', x.dtype), mine, lambda: _packed_modulated_weight(self, nst[:1], x.dtype, transposed=self.up == 2))
                    else:
                        x = gnerf_hip.scale_channels(x, nst)
                nxt = None
                if next_layer is not None and cl and _fast_path(x, next_w, next_layer.weight, next_layer.bias, next_layer.noise_strength, next_layer.affine.weight, next_layer.affine.bias):
                    nxt_aff = (next_layer.affine.weight, next_layer.affine.bias)
                    nxt = _per_latent(next_layer, next_w, 'nstyles', nxt_aff, lambda: gnerf_hip.normalise_styles(next_layer.affine(next_w)))
                    folded = True
                if own_plain and torgb_tail is not None and nxt is None and gnerf_hip.conv3x3_epilogue_torgb_supported(x, c_out):
                    # ... and the block's ToRGB, added to the running image, in the same launch; no layer output is written (round 6)
                    tg, w_rgb, img = torgb_tail
                    rgb_aff = (tg.affine.weight, tg.affine.bias)
                    rgb_styles = _per_latent(tg, w_rgb, 'styles', rgb_aff, lambda: tg.affine(w_rgb) * tg.weight_gain)
                    rgb_w = _per_latent(tg, w_rgb, 'rgbw16', rgb_aff + (tg.weight,), lambda: gnerf_hip.torgb_weights(tg.weight, rgb_styles))
                    gnerf_hip.conv3x3_epilogue_torgb(x, packed if packed is not None else _packed_prenormalised_weight(self, x.dtype), img, rgb_w, _cast_param(tg, 'bias', x.dtype),
                                                     tg.conv_clamp, bias=_cast_param(self, 'bias', x.dtype), scale=dco, noise=noise, round_noise=True,
                                                     gain=LRELU_GAIN * gain, clamp=clamp)
                    return _TORGB_DONE
                if own_plain:
                    # convolution + demodulation + noise + bias + lrelu + clamp (+ the next layer's input scaling) in one launch
                    x = gnerf_hip.conv3x3_epilogue(x, packed if packed is not None else _packed_prenormalised_weight(self, x.dtype), _cast_param(self, 'bias', x.dtype), scale=dco, noise=noise,
                                                   round_noise=True, gain=LRELU_GAIN * gain, clamp=clamp, next_scale=nxt)
                    return (x, folded) if next_layer is not None else x
                epi = dict(bias=_cast_param(self, 'bias', x.dtype), scale=dco, act='lrelu', gain=LRELU_GAIN * gain, clamp=clamp, next_scale=nxt) if noise is None else None
                if own_up:
                    w_t = ('phases', packed if packed is not None else _packed_prenormalised_weight(self, x.dtype, transposed=True))        # the x2 layer's transposed convolution on our own kernel
                else:
                    w_t = _prenormalised_weight(self, x.dtype, cl, transposed=True) if self.up == 2 else None
                out = self._resampled_conv(x, _prenormalised_weight(self, x.dtype, cl), 1, weight_t=w_t, epilogue=epi)
                x, done = out if epi is not None else (out, False)
                if not done:
                    if nxt is not None and not _is_channels_last(x):          # (the convolution gave back another layout: scale separately)
                        x = gnerf_hip.modconv_epilogue(x, _cast_param(self, 'bias', x.dtype), scale=dco, noise=noise, round_noise=True, act='lrelu', gain=LRELU_GAIN * gain, clamp=clamp)
                        x = gnerf_hip.scale_channels(x, nxt)
                    else:
                        x = gnerf_hip.modconv_epilogue(x, _cast_param(self, 'bias', x.dtype), scale=dco, noise=noise, round_noise=True, act='lrelu', gain=LRELU_GAIN * gain, clamp=clamp,
                                                       next_scale=nxt)
                return (x, folded) if next_layer is not None else x
            # per-sample weights in one launch, already in the order and memory format the convolution takes them
            # ([N,O,I,3,3] for conv2d, [N,I,O,3,3] for conv_transpose2d: re-ordering 38 MB of fp32 weights per up-layer was a
            # strided copy per call); constants of an orbit, like the styles
            wts = _per_latent(self, w, ('wts', x.dtype, cl), mine,
                              lambda: gnerf_hip.modulate_weights(self.weight, styles, True, out_dtype=x.dtype, transposed=self.up == 2, channels_last=cl)[0])
            wts = wts.reshape(-1, *wts.shape[2:]) if n > 1 else wts[0]
            epi = dict(bias=_cast_param(self, 'bias', x.dtype), act='lrelu', gain=LRELU_GAIN * gain, clamp=clamp) if (noise is None and n == 1) else None
            x = self._resampled_conv(x.reshape(1, n * c_in, h, wd) if n > 1 else x, None if self.up == 2 else wts, n, weight_t=wts if self.up == 2 else None,
                                     epilogue=epi)
            x, done = x if epi is not None else (x, False)
            x = x.reshape(n, c_out, *x.shape[2:]) if n > 1 else x
            if not done:
                x = gnerf_hip.modconv_epilogue(x, _cast_param(self, 'bias', x.dtype), noise=noise, act='lrelu', gain=LRELU_GAIN * gain, clamp=clamp)
            return (x, folded) if next_layer is not None else x
        # The reference's own flow (modulated_conv2d, networks_stylegan2.py:41-98, called from SynthesisLayer.forward :315-334 with
        # padding = kernel_size // 2 and flip_weight = (up == 1)): PyTorch ops for the modulation, the convolution through
        # torch_utils.ops.conv2d_resample and the demodulation through torch_utils.ops.fma -- the overlay's modules on a GPU, i.e. what
        # a G-NeRF checkout gets from this repo without any of the kernels above.
        if fused:
            wts = _modulated_weights(self.weight, styles, True, x.dtype == torch.float16).to(x.dtype)          # [N,O,I,3,3]
            c_out = wts.shape[1]
            x = conv2d_resample.conv2d_resample(x=x.reshape(1, n * c_in, h, wd), w=wts.reshape(n * c_out, c_in, 3, 3), f=self.resample_filter,
                                                up=self.up, padding=1, groups=n, flip_weight=(self.up == 1))
            x = x.reshape(n, c_out, *x.shape[2:])
            if noise is not None:
                x = x.add_(noise)
        else:
            weight = self.weight
            if x.dtype == torch.float16:
                weight, styles = _prenormalize(weight, styles)
            dcoefs = _demod_coefficients(weight, styles).to(x.dtype)[:, :, None, None]
            x = conv2d_resample.conv2d_resample(x=x * styles.to(x.dtype)[:, :, None, None], w=weight.to(x.dtype), f=self.resample_filter,
                                                up=self.up, padding=1, flip_weight=(self.up == 1))
            x = fma.fma(x, dcoefs, noise.to(x.dtype)) if noise is not None else x * dcoefs
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        x = bias_act.bias_act(x, self.bias.to(x.dtype), act='lrelu', gain=LRELU_GAIN * gain, clamp=clamp)
        return (x, folded) if next_layer is not None else x


class ToRGB(nn.Module):
    """ToRGBLayer (networks_stylegan2.py:349-367): modulated 1x1 convolution without demodulation, linear bias_act."""

    def __init__(self, c_in, c_out, w_dim, conv_clamp=None):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = Linear(w_dim, c_in, bias_init=1.0)
        self.weight = nn.Parameter(torch.randn(c_out, c_in, 1, 1))
        self.bias = nn.Parameter(torch.zeros(c_out))
        self.weight_gain = 1 / math.sqrt(c_in)

    def streams_into_image(self, x, w):
        """True when forward(x, w, ..., accumulate_into=img) adds the layer's output to the block's running image in its own launch."""
        import gnerf_hip
        return (_fast_path(x, w, self.weight, self.bias, self.affine.weight, self.affine.bias) and x.dtype == torch.float16 and _is_channels_last(x)
                and self.weight.shape[0] == 3 and x.shape[1] in gnerf_hip.TORGB_CHANNELS)

    def forward(self, x, w, fused=True, accumulate_into=None):
        n, c_in, h, wd = x.shape
        aff = (self.affine.weight, self.affine.bias)
        fast = _fast_path(x, w, self.weight, self.bias, *aff)
        styles = _per_latent(self, w, 'styles', aff, lambda: self.affine(w) * self.weight_gain) if fast else self.affine(w) * self.weight_gain
        assert accumulate_into is None or self.streams_into_image(x, w)
        if fast:
            import gnerf_hip
            if x.dtype == torch.float16 and _is_channels_last(x) and self.weight.shape[0] == 3 and c_in in gnerf_hip.TORGB_CHANNELS:
                return gnerf_hip.torgb_channels_last(x, self.weight, styles, _cast_param(self, 'bias', x.dtype), clamp=self.conv_clamp,
                                                     accumulate_into=accumulate_into)      # one streaming read of x
            if x.dtype == torch.float32 and _is_channels_last(x):
                # float32 channels_last (what the fp32-grade convolution hands over): the 1x1 modulated convolution is a GEMM on the tensor's own
                # memory, [H W, C] x [C, O] per sample with the styles folded into the small operand -- no scaling pass over x, no layout change,
                # and bit-reproducible (MIOpen's fp32 channels_last 1x1 kernels for 512 -> 96 and 256 -> 96 are not: they differ by an ulp from
                # call to call, tools/dbg_f32x3_det.py; the grouped per-sample form below would copy a batch back to NCHW for its reshape)
                # [O, C] x [C, H W] per sample, the activations as the TRANSPOSED operand (their channels_last memory is [H W, C]): the result is
                # a dense NCHW image, which is what it is added to (the block's running image, upsample2d_add_channels_last's `y`)
                wmod = _per_latent(self, w, 'wmod', aff + (self.weight,), lambda: (self.weight.reshape(1, -1, c_in) * styles[:, None, :]).contiguous())
                xt = x.permute(0, 2, 3, 1).reshape(n, h * wd, c_in).transpose(1, 2)
                if self.conv_clamp is None:
                    return torch.baddbmm(self.bias.reshape(1, -1, 1), wmod, xt).reshape(n, -1, h, wd)
                x = torch.bmm(wmod, xt).reshape(n, -1, h, wd)
            elif x.dtype == torch.float16 and n > 1:
                x = F.conv2d(gnerf_hip.scale_channels(x, styles), self.weight.to(x.dtype))
            else:
                wts = _per_latent(self, w, ('wts', x.dtype), aff + (self.weight,), lambda: gnerf_hip.modulate_weights(self.weight, styles, False, out_dtype=x.dtype)[0])
                x = F.conv2d(x.reshape(1, n * c_in, h, wd), wts.reshape(-1, c_in, 1, 1), groups=n).reshape(n, -1, h, wd)
            return gnerf_hip.modconv_epilogue(x, _cast_param(self, 'bias', x.dtype), act='linear', gain=1.0, clamp=self.conv_clamp)
        if fused:                       # (the reference's flow, as in StyledConv.forward)
            wts = _modulated_weights(self.weight, styles, False, False).to(x.dtype)
            x = conv2d_resample.conv2d_resample(x=x.reshape(1, n * c_in, h, wd), w=wts.reshape(-1, c_in, 1, 1), groups=n).reshape(n, -1, h, wd)
        else:
            x = conv2d_resample.conv2d_resample(x=x * styles.to(x.dtype)[:, :, None, None], w=self.weight.to(x.dtype))
        return bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)


class Block(nn.Module):
    """SynthesisBlock, 'skip' architecture (networks_stylegan2.py:371-470): [const | conv0 (x up)] -> conv1 -> ToRGB added to
    the (upsampled) running image."""

    def __init__(self, c_in, c_out, w_dim, resolution, img_channels, is_last, use_fp16=False, conv_clamp=None, up=2, emit_channels_last=False):
        super().__init__()
        self.c_in, self.up, self.use_fp16, self.is_last = c_in, up, use_fp16, is_last
        # The backbone's final block: its `upsample2d(img) + torgb(x)` IS the tri-plane image.  One fused kernel writes it
        # channels_last (memory [N,H,W,96]), the layout the fused renderer reads in place.
        self.emit_channels_last = emit_channels_last
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))
        if c_in == 0:
            self.const = nn.Parameter(torch.randn(c_out, resolution, resolution))
        else:
            self.conv0 = StyledConv(c_in, c_out, w_dim, resolution, up=up, conv_clamp=conv_clamp)
        self.conv1 = StyledConv(c_out, c_out, w_dim, resolution, conv_clamp=conv_clamp)
        self.torgb = ToRGB(c_out, img_channels, w_dim, conv_clamp=conv_clamp)
        self.num_conv, self.num_torgb = (1 if c_in == 0 else 2), 1

    def forward(self, x, img, ws, noise_mode='random', force_fp32=False, fused_modconv=None, discard_x=False):
        """discard_x: the caller reads only the image (the superresolution's last block): x may come back as None."""
        dtype = torch.float16 if self.use_fp16 and ws.is_cuda and not force_fp32 else torch.float32
        upsampled = False
        fused = (not self.training) if fused_modconv in (None, 'inference_only') else bool(fused_modconv)      # networks_stylegan2.py:433-434
        ws = ws.unbind(dim=1)
        if self.c_in == 0:
            x = self.const.to(dtype).unsqueeze(0).repeat(ws[0].shape[0], 1, 1, 1)
            x = self.conv1(x, ws[0], noise_mode, fused=fused)
        else:
            fast = _fast_path(x, ws[0], self.conv0.weight, self.conv1.weight, self.torgb.weight)
            # the reference's block entry fixes the layout as well as the type (networks_stylegan2.py:438, fp16_channels_last = False)
            x = x.to(dtype) if fast else x.to(dtype=dtype, memory_format=torch.contiguous_format)
            if dtype == torch.float16 and _FP16_CHANNELS_LAST and x.is_cuda and x.shape[1] % 8 == 0 and fast:
                x = x.contiguous(memory_format=torch.channels_last)
            x, folded = self.conv0(x, ws[0], noise_mode, fused=fused, next_layer=self.conv1, next_w=ws[1])
            tail = None
            if (discard_x and _FUSED_TORGB and img is not None and img.is_cuda and img.dtype == torch.float32 and not self.emit_channels_last
                    and self.torgb.streams_into_image(x, ws[-1]) and not torch.is_grad_enabled()):
                # the layer's convolution takes the ToRGB into its epilogue and adds the result to the (upsampled) running image: no x is written
                if self.up == 2:
                    img, upsampled = upfirdn2d.upsample2d(img, self.resample_filter), True
                if img.is_contiguous() and tuple(img.shape) == (x.shape[0], 3, x.shape[2], x.shape[3]):
                    tail = (self.torgb, ws[-1], img)
            x = self.conv1(x, ws[1], noise_mode, fused=fused, prescaled=folded, torgb_tail=tail)
            if x is _TORGB_DONE:
                return None, img
        if (img is not None and img.is_cuda and img.dtype == torch.float32 and not self.emit_channels_last and self.torgb.streams_into_image(x, ws[-1])
                and not (torch.is_grad_enabled() and img.requires_grad)):
            # upsample the running image, then let ToRGB add its output to it in its own launch (no fp16 y, no conversion, no add kernel)
            if self.up == 2 and not upsampled:
                img = upfirdn2d.upsample2d(img, self.resample_filter)
            if img.is_contiguous() and tuple(img.shape) == (x.shape[0], 3, x.shape[2], x.shape[3]):
                return x, self.torgb(x, ws[-1], fused=fused, accumulate_into=img)
            y = self.torgb(x, ws[-1], fused=fused).float()
            return x, img.add_(y)
        y = self.torgb(x, ws[-1], fused=fused).float()
        if img is not None and self.up == 2 and self.emit_channels_last and img.is_cuda:
            return x, upfirdn2d.upsample2d_add_channels_last(img, y.contiguous(), self.resample_filter)
        if img is not None and self.up == 2:
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if img is None:
            return x, y
        if torch.is_grad_enabled() and (img.requires_grad or y.requires_grad):
            # Out of place under autograd: the super-resolution's first block receives `img` as a VIEW of the feature image it
            # also convolves (triplane.py:86), and in fp32 the in-place form overwrites what that convolution saved for its
            # backward pass (the reference's own block64 raises here when trained without fp16; with fp16 its cast makes a copy).
            return x, img + y
        return x, img.add_(y)


class Synthesis(nn.Module):
    """SynthesisNetwork (networks_stylegan2.py:474-525): blocks b4 ... b<resolution>."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, emit_channels_last=False):
        super().__init__()
        self.block_resolutions = [2 ** i for i in range(2, int(math.log2(img_resolution)) + 1)]
        ch = {r: min(channel_base // r, channel_max) for r in self.block_resolutions}
        self.num_ws = 0
        for r in self.block_resolutions:
            blk = Block(ch[r // 2] if r > 4 else 0, ch[r], w_dim, r, img_channels, is_last=(r == img_resolution),
                        emit_channels_last=(r == img_resolution and emit_channels_last))
            self.num_ws += blk.num_conv + (blk.num_torgb if r == img_resolution else 0)
            setattr(self, f'b{r}', blk)

    def forward(self, ws, noise_mode='random', **block_kwargs):
        ws = ws.float()
        x = img = None
        i = 0
        for r in self.block_resolutions:
            blk = getattr(self, f'b{r}')
            x, img = blk(x, img, ws.narrow(1, i, blk.num_conv + blk.num_torgb), noise_mode, **block_kwargs)
            i += blk.num_conv
        return img


class Backbone(nn.Module):
    """networks_stylegan2.Generator (:529-557): mapping + synthesis to the 96-channel 256x256 plane image."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, num_mapping_layers=2, **synthesis_kwargs):
        super().__init__()
        self.synthesis = Synthesis(w_dim, img_resolution, img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = Mapping(z_dim, c_dim, w_dim, self.num_ws, num_layers=num_mapping_layers)


class SuperRes8XDC(nn.Module):
    """SuperresolutionHybrid8XDC (superresolution.py:266-303): 64^2 feature image -> block64 (no upsampling) -> antialiased
    bilinear resize to 128^2 -> block0 (256^2, 256 ch) -> block1 (512^2, 128 ch); fp16 with clamp 256 on the GPU."""

    def __init__(self, channels=32, w_dim=512, use_fp16=True, antialias=True):
        super().__init__()
        clamp = 256 if use_fp16 else None
        self.antialias = antialias
        self.block64 = Block(channels, channels, w_dim, 64, 3, is_last=True, use_fp16=use_fp16, conv_clamp=clamp, up=1)
        self.block0 = Block(channels, 256, w_dim, 256, 3, is_last=False, use_fp16=use_fp16, conv_clamp=clamp)
        self.block1 = Block(256, 128, w_dim, 512, 3, is_last=True, use_fp16=use_fp16, conv_clamp=clamp)

    def forward(self, rgb, x, ws, noise_mode='none', **block_kwargs):
        # (the same tensor OBJECT for the same ws: the layers below cache what depends on their latent slice by its identity)
        ws_in = ws
        ws = _per_latent(self, ws, 'ws3', (), lambda: ws[:, -1:, :].repeat(1, 3, 1))
        if _rows_share_latent(ws_in):
            _mark_one_latent(ws)
        x_raw, image_raw = self.block64(x, rgb, ws, noise_mode, **block_kwargs)
        if x.shape[-1] != 128:
            x = F.interpolate(x_raw, size=(128, 128), mode='bilinear', align_corners=False, antialias=self.antialias)
            rgb = F.interpolate(image_raw, size=(128, 128), mode='bilinear', align_corners=False, antialias=self.antialias)
        x, rgb = self.block0(x, rgb, ws, noise_mode, **block_kwargs)
        x, rgb = self.block1(x, rgb, ws, noise_mode, discard_x=True, **block_kwargs)
        return rgb, image_raw


FFHQ_RENDERING = {
    'depth_resolution': 48, 'depth_resolution_importance': 48, 'ray_start': 2.25, 'ray_end': 3.3, 'box_warp': 1,
    'disparity_space_sampling': False, 'clamp_mode': 'softplus', 'avg_camera_radius': 2.7, 'avg_camera_pivot': [0, 0, 0.2],
    'c_gen_conditioning_zero': False, 'c_scale': 1, 'superresolution_noise_mode': 'none', 'sr_antialias': True, 'decoder_lr_mul': 1,
}


class Generator(nn.Module):
    """TriPlaneGenerator (triplane.py:19-108), FFHQ configuration (train.py:238-377)."""

    def __init__(self, z_dim=512, c_dim=25, w_dim=512, rendering_kwargs=None, sr_use_fp16=True, planes_channels_last=True):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.rendering_kwargs = dict(FFHQ_RENDERING if rendering_kwargs is None else rendering_kwargs)
        self.renderer, self.ray_sampler = ImportanceRenderer(), RaySampler()
        self.backbone = Backbone(z_dim, c_dim, w_dim, img_resolution=256, img_channels=96, emit_channels_last=planes_channels_last)
        self.superresolution = SuperRes8XDC(32, w_dim, use_fp16=sr_use_fp16, antialias=self.rendering_kwargs.get('sr_antialias', True))
        self.decoder = H.TriPlaneDecoder(32, self.rendering_kwargs.get('decoder_lr_mul', 1), 32)
        self.neural_rendering_resolution = 64
        self._last_planes = None

    def mapping(self, z, c):
        if self.rendering_kwargs.get('c_gen_conditioning_zero', False):
            c = torch.zeros_like(c)
        return self.backbone.mapping(z, c * self.rendering_kwargs.get('c_scale', 0))

    def synthesis(self, ws, c, neural_rendering_resolution=None, update_emas=False, cache_backbone=False, use_cached_backbone=False,
                  only_depth=False, **synthesis_kwargs):
        """triplane.py:53-89.  synthesis_kwargs (noise_mode, force_fp32, fused_modconv) go to the backbone as given and to the
        superresolution without noise_mode, which the rendering options fix ('superresolution_noise_mode', triplane.py:87)."""
        res = self.neural_rendering_resolution = neural_rendering_resolution or self.neural_rendering_resolution
        o, d = self.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), res)
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            planes = self.backbone.synthesis(ws, **synthesis_kwargs)
        if cache_backbone:
            self._last_planes = planes
        n = planes.shape[0]
        if n == 1 and o.shape[0] > 1:
            # several cameras for ONE latent (an orbit's frames, gen_videos.py:150-166, batched): the renderer reads the one set of
            # planes for every view and gives each the results of a call of its own; the superresolution sees a batch of the latent
            ws = _per_latent(self, ws, ('views', o.shape[0]), (), lambda: ws.expand(o.shape[0], -1, -1).contiguous())
            _mark_one_latent(ws)
        feat, depth, _ = self.renderer(planes.view(n, 3, 32, *planes.shape[-2:]), self.decoder, o, d, self.rendering_kwargs)
        n = o.shape[0]
        feature_image = feat.permute(0, 2, 1).reshape(n, 32, res, res).contiguous()
        depth_image = depth.permute(0, 2, 1).reshape(n, 1, res, res)
        if only_depth:                                                                          # triplane.py:83-84
            return {'image': depth_image, 'image_raw': depth_image, 'image_depth': depth_image}
        sr_kwargs = {k: v for k, v in synthesis_kwargs.items() if k != 'noise_mode'}
        sr_image, raw = self.superresolution(feature_image[:, :3], feature_image, ws,
                                             noise_mode=self.rendering_kwargs.get('superresolution_noise_mode', 'none'), **sr_kwargs)
        return {'image': sr_image, 'image_raw': raw, 'image_depth': depth_image}

    def sample_mixed(self, coordinates, directions, ws, noise_mode='random', **_ignored):
        """Density / colour at arbitrary points for given latents (triplane.py:98-102; shape extraction, density regulariser)."""
        planes = self.backbone.synthesis(ws, noise_mode=noise_mode)
        return self.renderer.run_model(planes.view(len(planes), 3, 32, *planes.shape[-2:]), self.decoder, coordinates, directions, self.rendering_kwargs)

    def sample(self, coordinates, directions, z, c, **kw):
        return self.sample_mixed(coordinates, directions, self.mapping(z, c), **kw)

    def forward(self, z, c, **kw):
        return self.synthesis(self.mapping(z, c), c, **kw)


# ---------------------------------------------------------------------------------------------------------------------
# Discriminator on the 64x64 depth image (training_loop.py:183: Discriminator(c_dim=25, img_resolution=64, img_channels=1)).


class Conv(nn.Module):
    """Conv2dLayer (networks_stylegan2.py:140-197) for the cases the discriminator uses: 1x1 or 3x3, optional x2
    downsampling (conv2d_resample.py:96-111: a 1x1 kernel decimates through the low-pass filter first and convolves after,
    a 3x3 kernel blurs first and convolves with stride 2), bias + activation (+ clamp) in bias_act."""

    def __init__(self, c_in, c_out, kernel_size, bias=True, activation='linear', down=1, conv_clamp=None):
        super().__init__()
        self.activation, self.down, self.conv_clamp, self.padding = activation, down, conv_clamp, kernel_size // 2
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))
        self.weight_gain = 1 / math.sqrt(c_in * kernel_size ** 2)
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.weight = nn.Parameter(torch.randn(c_out, c_in, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(c_out)) if bias else None

    def forward(self, x, gain=1.0):
        w = (self.weight * self.weight_gain).to(x.dtype)
        if self.down == 1:
            x = F.conv2d(x, w, padding=self.padding)
        elif w.shape[-1] == 1:
            x = F.conv2d(upfirdn2d.upfirdn2d(x, self.resample_filter, down=2, padding=[1, 1, 1, 1]), w)
        else:
            p = self.padding + 1
            x = F.conv2d(upfirdn2d.upfirdn2d(x, self.resample_filter, padding=[p, p, p, p]), w, stride=2)
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        b = self.bias.to(x.dtype) if self.bias is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class DBlock(nn.Module):
    """DiscriminatorBlock, 'resnet' architecture (networks_stylegan2.py:561-651)."""

    def __init__(self, c_in, c_tmp, c_out, resolution, img_channels, use_fp16=False, conv_clamp=None):
        super().__init__()
        self.c_in, self.resolution, self.use_fp16 = c_in, resolution, use_fp16
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))
        if c_in == 0:
            self.fromrgb = Conv(img_channels, c_tmp, 1, activation='lrelu', conv_clamp=conv_clamp)
        self.conv0 = Conv(c_tmp, c_tmp, 3, activation='lrelu', conv_clamp=conv_clamp)
        self.conv1 = Conv(c_tmp, c_out, 3, activation='lrelu', down=2, conv_clamp=conv_clamp)
        self.skip = Conv(c_tmp, c_out, 1, bias=False, down=2)

    def forward(self, x, img, force_fp32=False):
        dtype = torch.float16 if self.use_fp16 and img.is_cuda and not force_fp32 else torch.float32
        if self.c_in == 0:
            x = self.fromrgb(img.to(dtype))
        else:
            x = x.to(dtype)
        y = self.skip(x, gain=math.sqrt(0.5))
        x = self.conv1(self.conv0(x), gain=math.sqrt(0.5))
        return y.add_(x)


def minibatch_std(x, group_size, num_channels=1):
    """MinibatchStdLayer (networks_stylegan2.py:655-683): one extra feature map per channel group holding the standard
    deviation over groups of `group_size` items, averaged over channels and pixels.  The batch must be a multiple of the group
    size (the reference's reshape fails otherwise -- train.py's default group of 3 with 4 items per GPU)."""
    n, c, h, w = x.shape
    g = min(group_size, n) if group_size is not None else n
    y = x.reshape(g, -1, num_channels, c // num_channels, h, w)
    y = (y - y.mean(dim=0)).square().mean(dim=0).add(1e-8).sqrt().mean(dim=[2, 3, 4])         # [n/g, F]
    return torch.cat([x, y.reshape(-1, num_channels, 1, 1).repeat(g, 1, h, w)], dim=1)


class DEpilogue(nn.Module):
    """DiscriminatorEpilogue (networks_stylegan2.py:687-744), 'resnet' architecture, conditioned by projection."""

    def __init__(self, c_in, cmap_dim, resolution, mbstd_group_size=4, mbstd_num_channels=1, conv_clamp=None):
        super().__init__()
        self.cmap_dim, self.mbstd_group_size, self.mbstd_num_channels = cmap_dim, mbstd_group_size, mbstd_num_channels
        self.conv = Conv(c_in + mbstd_num_channels, c_in, 3, activation='lrelu', conv_clamp=conv_clamp)
        self.fc = Linear(c_in * resolution ** 2, c_in, activation='lrelu')
        self.out = Linear(c_in, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, cmap):
        x = x.float()
        if self.mbstd_num_channels > 0:
            x = minibatch_std(x, self.mbstd_group_size, self.mbstd_num_channels)
        x = self.out(self.fc(self.conv(x).flatten(1)))
        if self.cmap_dim > 0:
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / math.sqrt(self.cmap_dim))
        return x


class LabelMapping(nn.Module):
    """MappingNetwork with z_dim = 0 and no broadcast (networks_stylegan2.py:200-272 as the discriminator builds it, :786):
    the camera label -> cmap_dim features through `embed` and eight lrelu layers."""

    def __init__(self, c_dim, w_dim, num_layers=8, lr_multiplier=0.01):
        super().__init__()
        self.num_layers = num_layers
        self.embed = Linear(c_dim, w_dim)
        for i in range(num_layers):
            setattr(self, f'fc{i}', Linear(w_dim, w_dim, activation='lrelu', lr_multiplier=lr_multiplier))

    def forward(self, c):
        x = _second_moment_normalize(self.embed(c.float()))
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        return x


class Discriminator(nn.Module):
    """networks_stylegan2.Discriminator (:748-799), 'resnet' blocks from img_resolution down to 8, epilogue at 4."""

    def __init__(self, c_dim=25, img_resolution=64, img_channels=1, channel_base=32768, channel_max=512, num_fp16_res=4,
                 conv_clamp=256, mbstd_group_size=4):
        super().__init__()
        self.c_dim = c_dim
        log2 = int(math.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(log2, 2, -1)]
        ch = {r: min(channel_base // r, channel_max) for r in self.block_resolutions + [4]}
        fp16_resolution = max(2 ** (log2 + 1 - num_fp16_res), 8)
        cmap_dim = ch[4] if c_dim > 0 else 0
        for r in self.block_resolutions:
            setattr(self, f'b{r}', DBlock(ch[r] if r < img_resolution else 0, ch[r], ch[r // 2], r, img_channels,
                                          use_fp16=(r >= fp16_resolution), conv_clamp=conv_clamp))
        if c_dim > 0:
            self.mapping = LabelMapping(c_dim, cmap_dim)
        self.b4 = DEpilogue(ch[4], cmap_dim, 4, mbstd_group_size=mbstd_group_size, conv_clamp=conv_clamp)

    def forward(self, img, c, update_emas=False, force_fp32=False):
        x = None
        for r in self.block_resolutions:
            x = getattr(self, f'b{r}')(x, img, force_fp32=force_fp32)
        return self.b4(x, self.mapping(c) if self.c_dim > 0 else None)
