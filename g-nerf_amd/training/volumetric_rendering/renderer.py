"""ImportanceRenderer with the reference's interface
(training/volumetric_rendering/renderer.py: generate_planes :23, project_onto_planes :39,
sample_from_planes :55, sample_from_3dgrid :67, ImportanceRenderer :82-253).

On a GPU the whole of `forward` -- depth proposals, tri-plane lookups, the decoder MLP, both ray
marches, importance resampling, the depth merge and the composite -- is ONE hand-written gfx950
kernel (csrc/render.hip through gnerf_hip.render_forward).  The random draws are still made by
torch, in the reference's order and shapes (rand_like([N,M,S,1]) then rand(N*M, F)), so a seeded
run consumes the generator exactly like the reference and produces the same image.

When autograd needs the gradient of the planes and/or the decoder (training), the same kernel runs
inside an autograd.Function whose backward is the second kernel (gnerf_hip.render_backward): it
recomputes the ray's forward pass and keeps nothing between the passes, where upstream autograd saves
every intermediate of the op chain (~3 GB at the training shape).

`run_model` (arbitrary points: sample / sample_mixed, incl. the density regulariser of training) has the
same pair of kernels (query_points / query_points_backward).

The PyTorch-op form below is what runs for CPU tensors (the reference's own behaviour: all of its
renderer is PyTorch ops), when the rays or points themselves need a gradient, or when `decoder` is
not the OSGDecoder 32->64->33 MLP.  A GPU call never silently degrades because the native library is absent:
gnerf_hip raises.
"""

import math
import warnings
import weakref

import torch
import torch.nn as nn

from training.volumetric_rendering.ray_marcher import MipRayMarcher2
from training.volumetric_rendering import math_utils

import os

import gnerf_hip

_KEEP_NHWC = os.environ.get('GNERF_KEEP_NHWC', '1') != '0'
# The two uniform draws made inside the render kernel (torch's own Philox stream, generator advanced as torch.rand would: same
# image, same generator state afterwards) instead of by two torch.rand launches.  Inference calls at 48+48 / 96+96 samples only.
_INKERNEL_RNG = os.environ.get('GNERF_INKERNEL_RNG', '0') == '1'
_NHWC_HINT = os.environ.get('GNERF_NHWC_PLANES', '0') == '1'


def generate_planes():
    """The three plane frames [3,3,3]; rows are the plane's axes.  With project_onto_planes they make
    plane 0 read (x,y), plane 1 (x,z) and plane 2 (z,x) -- EG3D's original choice, kept bit for bit."""
    return torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                         [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                         [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)


def project_onto_planes(planes, coordinates):
    """planes [P,3,3], coordinates [N,M,3] -> [N*P, M, 2]: coordinates expressed in each plane's frame,
    first two components."""
    N, M, _ = coordinates.shape
    P = planes.shape[0]
    pts = coordinates.unsqueeze(1).expand(-1, P, -1, -1).reshape(N * P, M, 3)
    frames = torch.linalg.inv(planes).unsqueeze(0).expand(N, -1, -1, -1).reshape(N * P, 3, 3)
    return torch.bmm(pts, frames)[..., :2]


def sample_from_planes(plane_axes, plane_features, coordinates, mode='bilinear', padding_mode='zeros', box_warp=None):
    """plane_features [N,P,C,H,W], coordinates [N,M,3] in world units -> [N,P,M,C]."""
    assert padding_mode == 'zeros'
    N, P, C, H, W = plane_features.shape
    M = coordinates.shape[1]
    grid = project_onto_planes(plane_axes, (2 / box_warp) * coordinates).unsqueeze(1)
    out = torch.nn.functional.grid_sample(plane_features.view(N * P, C, H, W), grid.float(), mode=mode,
                                          padding_mode=padding_mode, align_corners=False)
    return out.permute(0, 3, 2, 1).reshape(N, P, M, C)


def sample_from_3dgrid(grid, coordinates):
    """grid [1 or B,C,H,W,D], coordinates [B,P,3] -> [B,P,C] (trilinear, zero padding)."""
    B, P, nd = coordinates.shape
    out = torch.nn.functional.grid_sample(grid.expand(B, -1, -1, -1, -1), coordinates.reshape(B, 1, 1, -1, nd),
                                          mode='bilinear', padding_mode='zeros', align_corners=False)
    N, C, H, W, D = out.shape
    return out.permute(0, 4, 3, 2, 1).reshape(N, H * W * D, C)


def _interleaved_view(planes):
    """If `planes` [N,3,32,H,W] is a view of a channels_last [N,96,H,W] tensor -- memory [N,H,W,96], what a producer that writes
    channels_last hands over (torch_utils.ops.upfirdn2d.upsample2d_add_channels_last; triplane.py:74's view keeps the strides) --
    return that memory as a contiguous [N,H,W,96] view; else None.  The render kernels address this layout in place."""
    if planes.ndim != 5 or planes.dtype != torch.float32:
        return None
    N, P, C, H, W = planes.shape
    want = (P * C * H * W, C, 1, P * C * W, P * C)
    # (the stride of a size-1 dimension is arbitrary -- `view` of a one-item batch reports 96 for dim 0 -- and irrelevant)
    if N * P * C * H * W == 0 or any(sz > 1 and st != w for sz, st, w in zip(planes.shape, planes.stride(), want)):
        return None
    return planes.as_strided((N, H, W, P * C), (P * C * H * W, P * C * W, P * C, 1), planes.storage_offset())


def _planes_from_interleaved(g, like):
    """[N,H,W,96] gradient in the interleaved layout -> a [N,3,32,H,W] view of it (the strides of `like`)."""
    N, P, C, H, W = like.shape
    return g.view(N, H, W, P, C).permute(0, 3, 4, 1, 2)


def _osg_decoder_weights(decoder):
    """Return the effective (w1,b1,w2,b2) of an OSGDecoder-shaped module (triplane.py:113-122:
    FullyConnectedLayer(32,64) -> Softplus -> FullyConnectedLayer(64,33), both 'linear' with bias),
    or None if `decoder` is anything else."""
    net = getattr(decoder, 'net', None)
    if not isinstance(net, nn.Sequential) or len(net) != 3 or not isinstance(net[1], nn.Softplus):
        return None
    if net[1].beta != 1 or net[1].threshold != 20:
        return None
    fc1, fc2 = net[0], net[2]
    for fc in (fc1, fc2):
        if getattr(fc, 'activation', None) != 'linear' or getattr(fc, 'bias', None) is None or not hasattr(fc, 'weight_gain'):
            return None
    if tuple(fc1.weight.shape) != (64, 32) or tuple(fc2.weight.shape) != (33, 64):
        return None
    return fc1, fc2


class _FusedRender(torch.autograd.Function):
    """render_forward / render_backward as one differentiable op.  Inputs with a gradient: planes [N,3,32,H,W] and the
    decoder's effective weights; everything else is constant (importance depths are constants upstream as well,
    renderer.py:198/211).  `cfg` is the dict of static keyword arguments of gnerf_hip.render_forward."""

    @staticmethod
    def forward(ctx, planes, w1, b1, w2, b2, ray_origins, ray_dirs, noise_c, noise_f, ray_start, ray_end, cfg):
        N = planes.shape[0]
        nhwc = _interleaved_view(planes.detach())
        ctx.interleaved = nhwc is not None
        if ctx.interleaved:
            amax = _producer_absmax(planes)                 # None: the launcher measures it
        else:
            nhwc, amax = gnerf_hip.planes_to_nhwc(planes.detach().float(), with_absmax=True)
        out = gnerf_hip.render_forward(nhwc, N, (w1, b1, w2, b2), ray_origins, ray_dirs, noise_c, noise_f,
                                       ray_start=ray_start, ray_end=ray_end, planes_absmax=amax, **cfg)
        ctx.amax = amax if not ctx.interleaved else None     # (of the NHWC copy made here; a producer's tag is looked up again in backward)
        tensors = [planes, w1, b1, w2, b2, ray_origins, ray_dirs, noise_c]
        # The NHWC copy is kept for the backward pass (the planes' size again: 25 MB per item) unless GNERF_KEEP_NHWC=0, in
        # which case the backward pass repacks the saved NCHW planes a second time (37 us per 100 MB).
        ctx.nhwc = nhwc if (_KEEP_NHWC and not ctx.interleaved) else None
        ctx.has_fine = noise_f is not None
        ctx.limits_are_tensors = isinstance(ray_start, torch.Tensor)
        if ctx.has_fine:
            tensors.append(noise_f)
        if ctx.limits_are_tensors:
            tensors += [ray_start, ray_end]
        else:
            ctx.limits = (ray_start, ray_end)
        ctx.save_for_backward(*tensors)
        ctx.cfg = cfg
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rgb, g_depth, g_wsum):
        saved = list(ctx.saved_tensors)
        planes, w1, b1, w2, b2, ray_origins, ray_dirs, noise_c = saved[:8]
        rest = saved[8:]
        noise_f = rest.pop(0) if ctx.has_fine else None
        ray_start, ray_end = (rest[0], rest[1]) if ctx.limits_are_tensors else ctx.limits
        need_planes = ctx.needs_input_grad[0]
        need_decoder = any(ctx.needs_input_grad[1:5])
        N = planes.shape[0]
        if ctx.interleaved:
            nhwc = _interleaved_view(planes.detach())
            amax = _producer_absmax(planes)
        else:
            nhwc = ctx.nhwc if ctx.nhwc is not None else gnerf_hip.planes_to_nhwc(planes.detach().float())
            amax = ctx.amax
        g_planes, g_dec = gnerf_hip.render_backward(nhwc, N, (w1, b1, w2, b2), ray_origins, ray_dirs, noise_c, noise_f, g_rgb, g_depth, g_wsum,
                                                    ray_start=ray_start, ray_end=ray_end, need_planes=need_planes, need_decoder=need_decoder,
                                                    planes_absmax=amax, **ctx.cfg)
        grads = [None] * 12
        if need_planes:
            if ctx.interleaved:
                grads[0] = _planes_from_interleaved(g_planes, planes)                # laid out like the planes themselves: no repack
            else:
                grads[0] = gnerf_hip.planes_from_nhwc(g_planes, N).to(planes.dtype)     # contiguous NCHW, like the planes themselves
        if need_decoder:
            for i, (g, t) in enumerate(zip(g_dec, (w1, b1, w2, b2))):
                if ctx.needs_input_grad[1 + i]:
                    grads[1 + i] = g.to(t.dtype)
        return tuple(grads)


def _producer_absmax(planes):
    """max |planes| left on the producer's output by upsample2d_add_channels_last (valid while that tensor is unmodified)."""
    base = planes._base if planes._base is not None else planes
    tag = getattr(base, '_gnerf_absmax', None)
    if tag is not None and not base.is_inference() and tag[0] == base._version:
        return tag[1]
    return None


class _FusedQuery(torch.autograd.Function):
    """query_points / query_points_backward as one differentiable op (run_model for arbitrary points): gradients for the
    planes and the decoder's effective weights, none for the points."""

    @staticmethod
    def forward(ctx, planes, w1, b1, w2, b2, points, box_warp):
        nhwc = _interleaved_view(planes.detach())
        if nhwc is None:
            nhwc = gnerf_hip.planes_to_nhwc(planes.detach().float())
        sigma, rgb = gnerf_hip.query_points(nhwc, planes.shape[0], (w1, b1, w2, b2), points, box_warp)
        ctx.save_for_backward(planes, w1, b1, w2, b2, points)
        ctx.box_warp = box_warp
        return sigma, rgb

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_sigma, g_rgb):
        planes, w1, b1, w2, b2, points = ctx.saved_tensors
        need_planes = ctx.needs_input_grad[0]
        need_decoder = any(ctx.needs_input_grad[1:5])
        N = planes.shape[0]
        nhwc = _interleaved_view(planes.detach())
        interleaved = nhwc is not None
        if not interleaved:
            nhwc = gnerf_hip.planes_to_nhwc(planes.detach().float())
        g_planes, g_dec = gnerf_hip.query_points_backward(nhwc, N, (w1, b1, w2, b2), points, ctx.box_warp, g_sigma, g_rgb,
                                                          need_planes=need_planes, need_decoder=need_decoder)
        grads = [None] * 7
        if need_planes:
            grads[0] = _planes_from_interleaved(g_planes, planes) if interleaved else gnerf_hip.planes_from_nhwc(g_planes, N).to(planes.dtype)
        if need_decoder:
            for i, (g, t) in enumerate(zip(g_dec, (w1, b1, w2, b2))):
                if ctx.needs_input_grad[1 + i]:
                    grads[1 + i] = g.to(t.dtype)
        return tuple(grads)


_warned_fallbacks = set()


def _warn_gpu_fallback(reason, what='ImportanceRenderer.forward'):
    """One RuntimeWarning per (entry point, reason) and process when a call on GPU tensors runs the PyTorch-op form instead of the
    fused gfx950 kernels: correct, the reference's own arithmetic, but ~100x slower -- it must not happen unseen."""
    if (what, reason) not in _warned_fallbacks:
        _warned_fallbacks.add((what, reason))
        warnings.warn(f'{what}: GPU tensors, but {reason}: running the PyTorch-op form, not the fused HIP kernel', RuntimeWarning, stacklevel=3)


class ImportanceRenderer(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_marcher = MipRayMarcher2()
        self.plane_axes = generate_planes()

    # ------------------------------------------------------------------ dispatch

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options):
        """planes [N,3,32,H,W]; decoder(sampled_features [N,3,P,32], ray_directions) -> {'rgb','sigma'};
        rays [N,M,3]; rendering_options: the generator's rendering_kwargs dict (unknown keys ignored).
        Returns rgb [N,M,32], depth [N,M,1], weight_sum [N,M,1]."""
        self.plane_axes = self.plane_axes.to(ray_origins.device)
        # Extension (the reference requires one set of planes per item of rays): planes of ONE item with N > 1 items of rays are N
        # views of that object in one call, each with the results a call of its own would give -- its own pair of uniform draws, in
        # the order N calls would make them, and the final depth clamp (ray_marcher.py:49-50) over its own samples.
        views = planes.shape[0] == 1 and ray_origins.shape[0] > 1
        if planes.device.type == 'cuda':
            fcs = _osg_decoder_weights(decoder)
            rays_need_grad = torch.is_grad_enabled() and (ray_origins.requires_grad or ray_directions.requires_grad)
            needs_graph = torch.is_grad_enabled() and (planes.requires_grad or any(p.requires_grad for p in decoder.parameters()))
            noisy = rendering_options.get('density_noise', 0) != 0             # renderer.py:146-147: inside the kernels for forward calls (round 6)
            if fcs is not None and not rays_need_grad and planes.ndim == 5 and planes.shape[1] == 3 and planes.shape[2] == 32 \
                    and not (noisy and needs_graph):
                if views and needs_graph:
                    # the several-views launch is forward-only: with a graph, one differentiable call per view (same draws, same order)
                    outs = [self._forward_hip(planes, fcs, ray_origins[i:i + 1], ray_directions[i:i + 1], rendering_options, differentiable=True)
                            for i in range(ray_origins.shape[0])]
                    return tuple(torch.cat(t) for t in zip(*outs))
                return self._forward_hip(planes, fcs, ray_origins, ray_directions, rendering_options, differentiable=needs_graph)
            # A GPU call that leaves the fused kernel says so, once per reason (none of these occurs in gen_videos.py / train.py)
            _warn_gpu_fallback('the decoder is not the OSGDecoder 32->64->33 MLP' if fcs is None else
                               'the rays need a gradient' if rays_need_grad else
                               'density_noise > 0 (renderer.py:146-147) under autograd: the kernels take it for forward calls only' if noisy else
                               f'planes of shape {tuple(planes.shape)} are not [N,3,32,H,W]')
        if views:
            outs = [self._forward_torch(planes, decoder, ray_origins[i:i + 1], ray_directions[i:i + 1], rendering_options)
                    for i in range(ray_origins.shape[0])]
            return tuple(torch.cat(t) for t in zip(*outs))
        return self._forward_torch(planes, decoder, ray_origins, ray_directions, rendering_options)

    # ------------------------------------------------------------------ fused gfx950 path

    def _decoder_cache(self, fcs):
        fc1, fc2 = fcs
        params = (fc1.weight, fc1.bias, fc2.weight, fc2.bias)
        key = tuple((id(p), None if p.is_inference() else p._version, p.device) for p in params) + (float(fc1.weight_gain), float(fc1.bias_gain),
                                                                        float(fc2.weight_gain), float(fc2.bias_gain))
        cache = self.__dict__.get('_gnerf_decoder_cache')
        if cache is None or cache[0] != key or any(p.is_inference() for p in params):
            with torch.no_grad():
                eff = (fc1.weight.float() * fc1.weight_gain, fc1.bias.float() * fc1.bias_gain,        # networks_stylegan2.py:121-127
                       fc2.weight.float() * fc2.weight_gain, fc2.bias.float() * fc2.bias_gain)
                eff = tuple(t.contiguous() for t in eff)
            cache = (key, eff, params)      # keep params alive so ids stay unique
            self.__dict__['_gnerf_decoder_cache'] = cache
        return cache[1]

    def _planes_nhwc(self, planes):
        """(NHWC copy, max |planes|) of `planes`, cached on the tensor's identity and version: an orbit with cached backbone
        planes (triplane.py:66-71) converts once.  Inference tensors carry no version counter: they are converted every call.
        The cached copy is only handed to work on the stream that made it (another stream converts again, for itself)."""
        inter = _interleaved_view(planes.detach() if not planes.is_inference() else planes)
        if inter is not None:                               # already the renderer's layout (a channels_last producer): no copy
            amax = _producer_absmax(planes)
            if amax is not None or planes.is_inference():
                return inter, amax                          # (None: the launcher measures max |planes| itself)
        elif _NHWC_HINT and planes.ndim == 5 and planes.shape[1] == 3:
            from torch_utils.ops import upfirdn2d           # opt-in: ask the producer for channels_last planes from now on
            upfirdn2d.channels_last_output_shapes.add((planes.shape[1] * planes.shape[2], planes.shape[3], planes.shape[4]))
        if planes.is_inference():
            return gnerf_hip.planes_to_nhwc(planes.float(), with_absmax=True)
        base = planes._base if planes._base is not None else planes
        stream = torch.cuda.current_stream(planes.device).cuda_stream
        cache = self.__dict__.get('_gnerf_planes_cache')
        if cache is not None:
            ref, version, ptr, shape, made_on, out = cache
            if ref() is base and version == base._version and ptr == planes.data_ptr() and shape == tuple(planes.shape) \
                    and (made_on is None or made_on == stream):
                return out
        if inter is not None:
            out = (inter, gnerf_hip.planes_absmax(inter))   # measured once per planes tensor and version, like the repack
        else:
            out = gnerf_hip.planes_to_nhwc(planes.detach().float(), with_absmax=True)
        self.__dict__['_gnerf_planes_cache'] = (weakref.ref(base), base._version, planes.data_ptr(), tuple(planes.shape), stream, out)
        return out

    def pin_planes(self, planes):
        """Convert `planes` [N,3,32,H,W] now, wait for the device, and make the copy valid on EVERY stream (a HIP-graph
        capture runs on its own stream).  Returns an opaque handle that keeps the converted tensors alive: a captured
        graph bakes in their addresses, so whoever replays it must hold the handle and call `repin(handle)` before a replay
        if other planes went through this renderer in between (gen_videos_mi355x.FrameProgram does)."""
        base = planes._base if planes._base is not None else planes
        inter = _interleaved_view(planes.detach())
        if inter is not None:
            amax = _producer_absmax(planes)
            out = (inter, gnerf_hip.planes_absmax(inter) if amax is None else amax)
        else:
            out = gnerf_hip.planes_to_nhwc(planes.detach().float(), with_absmax=True)
        torch.cuda.synchronize(planes.device)
        handle = (weakref.ref(base), base._version, planes.data_ptr(), tuple(planes.shape), None, out)
        self.__dict__['_gnerf_planes_cache'] = handle
        return handle

    def repin(self, handle):
        self.__dict__['_gnerf_planes_cache'] = handle

    def _forward_hip(self, planes, fcs, ray_origins, ray_directions, opts, differentiable=False):
        N, M, _ = ray_origins.shape
        S = int(opts['depth_resolution'])
        F = int(opts['depth_resolution_importance'])
        if S > gnerf_hip.MAX_SAMPLES or F > gnerf_hip.MAX_SAMPLES:
            raise RuntimeError(f'ImportanceRenderer: at most {gnerf_hip.MAX_SAMPLES} coarse and fine samples per ray are supported')
        dev = ray_origins.device
        views = planes.shape[0] == 1 and N > 1              # N views of one item's planes (see forward)
        assert not (views and differentiable)               # forward() makes one differentiable call per view

        def limits(o, d):
            ray_start, ray_end = math_utils.get_ray_limits_box(o, d, box_side_length=opts['box_warp'])
            ok = ray_end > ray_start
            if torch.any(ok).item():                                     # renderer.py:94-96
                ray_start[~ok] = ray_start[ok].min()
                ray_end[~ok] = ray_start[ok].max()
            return ray_start, ray_end
        if opts['ray_start'] == opts['ray_end'] == 'auto':
            if views:                                                    # the fill-in values above are per call: per view here
                ray_start, ray_end = (torch.cat(t) for t in zip(*[limits(ray_origins[i:i + 1], ray_directions[i:i + 1]) for i in range(N)]))
            else:
                ray_start, ray_end = limits(ray_origins, ray_directions)
        else:
            ray_start, ray_end = opts['ray_start'], opts['ray_end']
        side = math.isqrt(M)
        if opts['clamp_mode'] != 'softplus':
            assert False, "MipRayMarcher only supports `clamp_mode`=`softplus`!"
        cfg = dict(depth_resolution=S, depth_resolution_importance=F, box_warp=opts['box_warp'],
                   white_back=bool(opts.get('white_back', False)), disparity_space_sampling=bool(opts.get('disparity_space_sampling', False)),
                   image_width=side if side * side == M else 0)
        dn = float(opts.get('density_noise', 0) or 0)
        assert not (dn and differentiable)                   # forward() sends that combination to the PyTorch-op form
        if _INKERNEL_RNG and not differentiable and not dn and gnerf_hip.render_generated_supported(S, F, ray_start, ray_end, cfg['disparity_space_sampling']) \
                and not torch.cuda.is_current_stream_capturing():
            # the same two draws (per view: the draws of a call of its own), made by the render kernel from the generator's state.  The
            # generator moves only once the launch has been accepted: a geometry the kernel does not reproduce (a device whose CU count
            # makes ATen's grid neither a power of two nor as large as the draw), an environment override of the kernel choice or planes
            # beyond 4 GB leave it where it was, and the call draws tensors below like any other.
            plan = gnerf_hip.torch_philox_plan(dev, N, M, S, F, per_item=views, advance=False)
            nhwc, amax = self._planes_nhwc(planes)
            if gnerf_hip.render_generated_supported(S, F, ray_start, ray_end, cfg['disparity_space_sampling'], plan=plan, numel_planes=nhwc.numel() // (1 if views else N)):
                try:
                    out = gnerf_hip.render_forward(nhwc, N, self._decoder_cache(fcs), ray_origins.detach(), ray_directions.detach(), None, None,
                                                   ray_start=ray_start, ray_end=ray_end, planes_absmax=amax, planes_shared=views, depth_clamp_per_item=views,
                                                   rng=plan, **cfg)
                except RuntimeError as e:
                    if 'failed (-3)' not in str(e):              # GNERF_E_UNSUPPORTED: the tensor form below; anything else is an error
                        raise
                else:
                    gnerf_hip.commit_philox_plan(plan)
                    return out
        # the reference's two draws, same shapes, same order (renderer.py:176/186/190 then :241)
        # ... and with density_noise the two normal draws of run_model between and behind them: rand_like, randn_like (coarse), rand, randn_like (fine)
        def draw(n):
            c = torch.rand([n, M, S, 1], device=dev, dtype=torch.float32)
            sc = torch.randn([n, M * S, 1], device=dev, dtype=torch.float32) if dn else None
            f = torch.rand(n * M, F, device=dev) if F > 0 else None
            sf = torch.randn([n, M * F, 1], device=dev, dtype=torch.float32) if (dn and F > 0) else None
            return c, sc, f, sf
        if views:
            draws = [draw(1) for _ in range(N)]
            noise_c, sig_c, noise_f, sig_f = (torch.cat(t) if t[0] is not None else None for t in zip(*draws))
        else:
            noise_c, sig_c, noise_f, sig_f = draw(N)
        sigma_noise = (sig_c.reshape(N * M, S) * dn, None if sig_f is None else sig_f.reshape(N * M, F) * dn) if dn else None
        if differentiable:
            fc1, fc2 = fcs
            eff = (fc1.weight.float() * fc1.weight_gain, fc1.bias.float() * fc1.bias_gain,          # networks_stylegan2.py:121-127
                   fc2.weight.float() * fc2.weight_gain, fc2.bias.float() * fc2.bias_gain)
            return _FusedRender.apply(planes, *eff, ray_origins.detach(), ray_directions.detach(), noise_c.reshape(N * M, S), noise_f,
                                      ray_start, ray_end, cfg)
        nhwc, amax = self._planes_nhwc(planes)
        return gnerf_hip.render_forward(nhwc, N, self._decoder_cache(fcs), ray_origins.detach(), ray_directions.detach(),
                                        noise_c, noise_f, ray_start=ray_start, ray_end=ray_end, planes_absmax=amax,
                                        planes_shared=views, depth_clamp_per_item=views, sigma_noise=sigma_noise, **cfg)

    # ------------------------------------------------------------------ PyTorch-op path

    def _forward_torch(self, planes, decoder, ray_origins, ray_directions, rendering_options):
        if rendering_options['ray_start'] == rendering_options['ray_end'] == 'auto':
            ray_start, ray_end = math_utils.get_ray_limits_box(ray_origins, ray_directions, box_side_length=rendering_options['box_warp'])
            ok = ray_end > ray_start
            if torch.any(ok).item():
                ray_start[~ok] = ray_start[ok].min()
                ray_end[~ok] = ray_start[ok].max()
        else:
            ray_start, ray_end = rendering_options['ray_start'], rendering_options['ray_end']
        depths_coarse = self.sample_stratified(ray_origins, ray_start, ray_end, rendering_options['depth_resolution'],
                                               rendering_options['disparity_space_sampling'])
        N, M, S, _ = depths_coarse.shape

        def shade(depths):
            count = depths.shape[2]
            pts = (ray_origins.unsqueeze(-2) + depths * ray_directions.unsqueeze(-2)).reshape(N, -1, 3)
            dirs = ray_directions.unsqueeze(-2).expand(-1, -1, count, -1).reshape(N, -1, 3)
            out = self._run_model(planes, decoder, pts, dirs, rendering_options, warn=False)        # (forward() has said why already)
            return out['rgb'].reshape(N, M, count, out['rgb'].shape[-1]), out['sigma'].reshape(N, M, count, 1)

        colors_coarse, densities_coarse = shade(depths_coarse)
        n_fine = rendering_options['depth_resolution_importance']
        if n_fine > 0:
            _, _, weights = self.ray_marcher(colors_coarse, densities_coarse, depths_coarse, rendering_options)
            depths_fine = self.sample_importance(depths_coarse, weights, n_fine)
            colors_fine, densities_fine = shade(depths_fine)
            all_depths, all_colors, all_densities = self.unify_samples(depths_coarse, colors_coarse, densities_coarse,
                                                                       depths_fine, colors_fine, densities_fine)
            rgb_final, depth_final, weights = self.ray_marcher(all_colors, all_densities, all_depths, rendering_options)
        else:
            rgb_final, depth_final, weights = self.ray_marcher(colors_coarse, densities_coarse, depths_coarse, rendering_options)
        return rgb_final, depth_final, weights.sum(2)

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        """Decoder outputs at arbitrary points [N,P,3] -> {'rgb' [N,P,32], 'sigma' [N,P,1]}
        (entry point of TriPlaneGenerator.sample / sample_mixed, triplane.py:91-102)."""
        return self._run_model(planes, decoder, sample_coordinates, sample_directions, options)

    def _run_model(self, planes, decoder, sample_coordinates, sample_directions, options, warn=True):
        self.plane_axes = self.plane_axes.to(sample_coordinates.device)
        density_noise = options.get('density_noise', 0)
        if planes.device.type == 'cuda' and planes.ndim == 5 and planes.shape[1] == 3 and planes.shape[2] == 32:
            fcs = _osg_decoder_weights(decoder)
            points_need_grad = torch.is_grad_enabled() and sample_coordinates.requires_grad
            if fcs is not None and not points_need_grad:
                needs_graph = torch.is_grad_enabled() and (planes.requires_grad or any(p.requires_grad for p in decoder.parameters()))
                if needs_graph:
                    fc1, fc2 = fcs
                    eff = (fc1.weight.float() * fc1.weight_gain, fc1.bias.float() * fc1.bias_gain,      # networks_stylegan2.py:121-127
                           fc2.weight.float() * fc2.weight_gain, fc2.bias.float() * fc2.bias_gain)
                    sigma, rgb = _FusedQuery.apply(planes, *eff, sample_coordinates.detach(), options['box_warp'])
                else:
                    sigma, rgb = gnerf_hip.query_points(self._planes_nhwc(planes)[0], planes.shape[0], self._decoder_cache(fcs),
                                                        sample_coordinates.detach(), options['box_warp'])
                out = {'rgb': rgb, 'sigma': sigma}
                if density_noise > 0:
                    out['sigma'] = out['sigma'] + torch.randn_like(out['sigma']) * density_noise
                return out
        if warn and planes.device.type == 'cuda':
            _warn_gpu_fallback('the points need a gradient, the decoder is not the OSGDecoder MLP or the planes are not [N,3,32,H,W]', 'ImportanceRenderer.run_model')
        feats = sample_from_planes(self.plane_axes, planes, sample_coordinates, padding_mode='zeros', box_warp=options['box_warp'])
        out = decoder(feats, sample_directions)
        if density_noise > 0:
            out['sigma'] += torch.randn_like(out['sigma']) * density_noise
        return out

    def query_sigma(self, planes, decoder, sample_coordinates, options):
        """Densities [N,P,1] only -- run_model(...)['sigma'] without evaluating or writing the 32 colour channels (not part of the
        reference's interface: used by this repo's shape-extraction harness, where rgb would be 17 GB of unread output at 512^3)."""
        if planes.device.type == 'cuda' and planes.ndim == 5 and planes.shape[1] == 3 and planes.shape[2] == 32 and not torch.is_grad_enabled():
            fcs = _osg_decoder_weights(decoder)
            if fcs is not None and options.get('density_noise', 0) == 0:
                return gnerf_hip.query_points(self._planes_nhwc(planes)[0], planes.shape[0], self._decoder_cache(fcs),
                                              sample_coordinates.detach(), options['box_warp'], want_rgb=False)[0]
        dirs = torch.zeros_like(sample_coordinates)
        dirs[..., -1] = -1
        return self.run_model(planes, decoder, sample_coordinates, dirs, options)['sigma']

    def sort_samples(self, all_depths, all_colors, all_densities):
        _, order = torch.sort(all_depths, dim=-2)
        all_depths = torch.gather(all_depths, -2, order)
        all_colors = torch.gather(all_colors, -2, order.expand(-1, -1, -1, all_colors.shape[-1]))
        all_densities = torch.gather(all_densities, -2, order.expand(-1, -1, -1, 1))
        return all_depths, all_colors, all_densities

    def unify_samples(self, depths1, colors1, densities1, depths2, colors2, densities2):
        return self.sort_samples(torch.cat([depths1, depths2], dim=-2), torch.cat([colors1, colors2], dim=-2),
                                 torch.cat([densities1, densities2], dim=-2))

    def sample_stratified(self, ray_origins, ray_start, ray_end, depth_resolution, disparity_space_sampling=False):
        """Jittered, roughly uniform depths [N,M,S,1] along each ray."""
        N, M, _ = ray_origins.shape
        dev = ray_origins.device
        if disparity_space_sampling:
            d = torch.linspace(0, 1, depth_resolution, device=dev).reshape(1, 1, depth_resolution, 1).repeat(N, M, 1, 1)
            d += torch.rand_like(d) * (1 / (depth_resolution - 1))
            return 1. / (1. / ray_start * (1. - d) + 1. / ray_end * d)
        if type(ray_start) == torch.Tensor:
            depths = math_utils.linspace(ray_start, ray_end, depth_resolution).permute(1, 2, 0, 3)
            step = (ray_end - ray_start) / (depth_resolution - 1)
            depths += torch.rand_like(depths) * step[..., None]
            return depths
        depths = torch.linspace(ray_start, ray_end, depth_resolution, device=dev).reshape(1, 1, depth_resolution, 1).repeat(N, M, 1, 1)
        depths += torch.rand_like(depths) * ((ray_end - ray_start) / (depth_resolution - 1))
        return depths

    def sample_importance(self, z_vals, weights, N_importance):
        """Depths [N,M,F,1] drawn from the smoothed coarse weights (NeRF-style hierarchical sampling)."""
        with torch.no_grad():
            N, M, S, _ = z_vals.shape
            z = z_vals.reshape(N * M, S)
            w = weights.reshape(N * M, -1)          # S-1 interval weights
            w = torch.nn.functional.max_pool1d(w.unsqueeze(1).float(), 2, 1, padding=1)
            w = torch.nn.functional.avg_pool1d(w, 2, 1).squeeze() + 0.01
            mids = 0.5 * (z[:, :-1] + z[:, 1:])
            fine = self.sample_pdf(mids, w[:, 1:-1], N_importance).detach()
        return fine.reshape(N, M, N_importance, 1)

    def sample_pdf(self, bins, weights, N_importance, det=False, eps=1e-5):
        """Inverse-transform sampling: bins [R,K+1], weights [R,K] -> samples [R,N_importance]."""
        R, K = weights.shape
        weights = weights + eps
        pdf = weights / torch.sum(weights, -1, keepdim=True)
        cdf = torch.cumsum(pdf, -1)
        cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
        if det:
            u = torch.linspace(0, 1, N_importance, device=bins.device).expand(R, N_importance)
        else:
            u = torch.rand(R, N_importance, device=bins.device)
        u = u.contiguous()
        idx = torch.searchsorted(cdf, u, right=True)
        lo = torch.clamp_min(idx - 1, 0)
        hi = torch.clamp_max(idx, K)
        pair = torch.stack([lo, hi], -1).view(R, 2 * N_importance)
        cdf_g = torch.gather(cdf, 1, pair).view(R, N_importance, 2)
        bins_g = torch.gather(bins, 1, pair).view(R, N_importance, 2)
        denom = cdf_g[..., 1] - cdf_g[..., 0]
        denom[denom < eps] = 1          # empty bin: never drawn, any value works
        return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])
