#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the fixtures it writes
are plain data (inputs + the reference's outputs) and are what travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference draws its two noise tensors from the global torch RNG
(renderer.py:176/186/190 `torch.rand_like`, renderer.py:241 `torch.rand`); we
record those draws so that any other implementation can replay them.
"""

import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference/g_nerf'
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.path.insert(0, REF)
    # networks_stylegan2.py:30 imports torchvision (absent here) for the ResNeXt encoder only.
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvr = types.ModuleType('torchvision.models.resnet')
    tvr.ResNet = type('ResNet', (torch.nn.Module,), {})
    tvr.Bottleneck = type('Bottleneck', (torch.nn.Module,), {})
    tvm.resnet = tvr
    tv.models = tvm
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm, 'torchvision.models.resnet': tvr})


class RecordNoise:
    """Context manager recording every torch.rand_like / torch.rand result (and, apart, every torch.randn_like one: the density noise
    of renderer.py:146-147)."""

    def __enter__(self):
        self.draws, self.normal = [], []
        self._rl, self._r, self._nl = torch.rand_like, torch.rand, torch.randn_like

        def randn_like(*a, **k):
            t = self._nl(*a, **k)
            self.normal.append(t.clone())
            return t
        torch.randn_like = randn_like

        def rand_like(*a, **k):
            t = self._rl(*a, **k)
            self.draws.append(t.clone())
            return t

        def rand(*a, **k):
            t = self._r(*a, **k)
            self.draws.append(t.clone())
            return t

        torch.rand_like, torch.rand = rand_like, rand
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand, torch.randn_like = self._rl, self._r, self._nl


def np_(t):
    return t.detach().cpu().numpy()


def make_render_case(name, seed, N, res, S, F, plane_hw, plane_scale, extra_opts=None, lr_mul=1.0):
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from training.volumetric_rendering.ray_sampler import RaySampler
    from training.triplane import OSGDecoder
    from camera_utils import LookAtPoseSampler

    torch.manual_seed(seed)
    H, W = plane_hw
    planes = torch.randn(N, 3, 32, H, W) * plane_scale
    dec = OSGDecoder(32, {'decoder_lr_mul': lr_mul, 'decoder_output_dim': 32})
    with torch.no_grad():
        dec.net[0].bias.normal_(0, 0.3)
        dec.net[2].bias.normal_(0, 0.3)
    cams = []
    for n in range(N):
        cams.append(LookAtPoseSampler.sample(3.14 / 2 + 0.4 * n - 0.2, 3.14 / 2 - 0.05 + 0.1 * n, radius=2.7))
    cam2world = torch.cat(cams, 0)
    intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).unsqueeze(0).repeat(N, 1, 1)
    origins, dirs = RaySampler()(cam2world, intr, res)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1,
                clamp_mode='softplus', disparity_space_sampling=False, white_back=False)
    if extra_opts:
        opts.update(extra_opts)
    ren = ImportanceRenderer()
    captured = {}
    orig_sample_importance = ren.sample_importance
    orig_run_model = ren.run_model
    orig_marcher = ren.ray_marcher.forward

    def sample_importance(z, w, n_imp):
        out = orig_sample_importance(z, w, n_imp)
        captured['weights_coarse'] = w.clone()
        captured['depths_fine'] = out.clone()
        return out

    calls = []

    def run_model(*a, **k):
        out = orig_run_model(*a, **k)
        calls.append({k2: v.clone() for k2, v in out.items()})
        return out

    def marcher(colors, densities, depths, options):
        captured.setdefault('march_depths', []).append(depths.clone())
        return orig_marcher(colors, densities, depths, options)

    ren.sample_importance = sample_importance
    ren.run_model = run_model
    ren.ray_marcher.forward = marcher
    with torch.no_grad(), RecordNoise() as rec:
        rgb, depth, wsum = ren(planes, dec, origins, dirs, opts)
    assert len(rec.draws) == (2 if F > 0 else 1)
    out = dict(
        planes=np_(planes), cam2world=np_(cam2world), intrinsics=np_(intr), res=np.int32(res),
        ray_origins=np_(origins), ray_dirs=np_(dirs),
        w1=np_(dec.net[0].weight), b1=np_(dec.net[0].bias), w2=np_(dec.net[2].weight), b2=np_(dec.net[2].bias),
        lr_mul=np.float32(lr_mul),
        noise_coarse=np_(rec.draws[0]).reshape(N, res * res, S),
        depth_resolution=np.int32(S), depth_resolution_importance=np.int32(F),
        ray_start=np.float32(opts['ray_start']), ray_end=np.float32(opts['ray_end']), box_warp=np.float32(opts['box_warp']),
        white_back=np.int32(opts['white_back']), disparity=np.int32(opts['disparity_space_sampling']),
        out_rgb=np_(rgb), out_depth=np_(depth), out_wsum=np_(wsum),
        sigma_coarse=np_(calls[0]['sigma']).reshape(N, res * res, S),
        depths_coarse=np_(captured['march_depths'][0]).reshape(N, res * res, S),
    )
    if opts.get('density_noise', 0) > 0:
        # the two randn_like draws of run_model, in call order (coarse pass, fine pass), as the reference ADDS them: times density_noise
        assert len(rec.normal) == (2 if F > 0 else 1)
        out.update(density_noise=np.float32(opts['density_noise']), sigma_noise_coarse=np_(rec.normal[0] * opts['density_noise']).reshape(N * res * res, S))
        if F > 0:
            out.update(sigma_noise_fine=np_(rec.normal[1] * opts['density_noise']).reshape(N * res * res, F))
    if F > 0:
        out.update(
            noise_fine=np_(rec.draws[1]),
            weights_coarse=np_(captured['weights_coarse']).reshape(N, res * res, S - 1),
            depths_fine=np_(captured['depths_fine']).reshape(N, res * res, F),
            sigma_fine=np_(calls[1]['sigma']).reshape(N, res * res, F),
            depths_all=np_(captured['march_depths'][1]).reshape(N, res * res, S + F),
        )
    np.savez(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim > 0})


def make_stage_cases():
    from training.volumetric_rendering.renderer import ImportanceRenderer, project_onto_planes, generate_planes, sample_from_planes
    from training.volumetric_rendering.ray_marcher import MipRayMarcher2
    torch.manual_seed(11)
    ren = ImportanceRenderer()
    out = {}
    # sample_pdf edge cases: uniform, spiky, all-zero-ish weights (renderer.py:248-249 denom rule)
    R, S, F = 6, 12, 10
    z = torch.sort(torch.rand(R, S) * 1.05 + 2.25, dim=1)[0]
    w = torch.rand(R, S - 1)
    w[1] = 0.0
    w[2] = 0.0
    w[2, 5] = 1.0
    w[3, :4] = 0.0
    w[4] = 1e-12
    with RecordNoise() as rec:
        fine = ren.sample_importance(z.reshape(1, R, S, 1), w.reshape(1, R, S - 1, 1), F)
    out.update(pdf_depths=np_(z), pdf_weights=np_(w), pdf_noise=np_(rec.draws[0]), pdf_out=np_(fine).reshape(R, F))
    # ray marcher incl. strongly negative densities (tiny weights) and white_back
    m = MipRayMarcher2()
    R, S, C = 5, 9, 4
    col = torch.rand(1, R, S, C)
    sig = torch.randn(1, R, S, 1) * 4
    sig[0, 1] = -200.0        # weights underflow to exactly 0 -> depth NaN -> inf -> clamp to global max
    sig[0, 2] = 60.0
    dep = torch.sort(torch.rand(1, R, S, 1) * 1.05 + 2.25, dim=2)[0]
    for wb in (0, 1):
        rgb, depth, wts = m(col, sig, dep, {'clamp_mode': 'softplus', 'white_back': bool(wb)})
        out.update({f'march_rgb_wb{wb}': np_(rgb), f'march_depth_wb{wb}': np_(depth), f'march_w_wb{wb}': np_(wts)})
    out.update(march_colors=np_(col), march_sigma=np_(sig), march_depths=np_(dep))
    # plane projection + bilinear lookup incl. out-of-range points and H != W
    planes = torch.randn(1, 3, 5, 6, 7)
    pts = (torch.rand(1, 40, 3) - 0.5) * 1.4
    uv = project_onto_planes(generate_planes(), pts * 2.0)
    feats = sample_from_planes(generate_planes(), planes, pts, padding_mode='zeros', box_warp=1)
    out.update(proj_points=np_(pts), proj_uv=np_(uv), lookup_planes=np_(planes), lookup_out=np_(feats))
    # ray/box limits ('auto' ray_start/ray_end, math_utils.py:47-98): hits, misses, axis-parallel rays
    from training.volumetric_rendering import math_utils
    bo = (torch.rand(1, 30, 3) - 0.5) * 4
    bd = torch.nn.functional.normalize(torch.randn(1, 30, 3), dim=-1)
    bd[0, :10] = torch.nn.functional.normalize(-bo[0, :10] + 0.2 * torch.randn(10, 3), dim=-1)   # aimed at the box
    bd[0, 10] = torch.tensor([0., 0., 1.]); bo[0, 10] = torch.tensor([0.1, -0.2, -3.])
    bd[0, 11] = torch.tensor([1., 0., 0.]); bo[0, 11] = torch.tensor([-3., 0.7, 0.])
    tmin, tmax = math_utils.get_ray_limits_box(bo, bd, box_side_length=1.0)
    out.update(box_origins=np_(bo), box_dirs=np_(bd), box_tmin=np_(tmin), box_tmax=np_(tmax))
    np.savez(os.path.join(HERE, 'stages.npz'), **out)
    print('stages.npz', sorted(out))


def make_camera_cases():
    from camera_utils import LookAtPoseSampler
    from training.volumetric_rendering.ray_sampler import RaySampler
    out = {}
    frame_num = 120
    idx = [0, 30, 60, 119]
    mats = []
    for i in idx:   # gen_videos.py:155-158
        mats.append(LookAtPoseSampler.sample(3.14 / 2 + 0.7 * np.sin(2 * 3.14 * i / frame_num),
                                             3.14 / 2 - 0.05 + 0.3 * np.cos(2 * 3.14 * i / frame_num), radius=2.7))
    out['orbit_frames'] = np.array(idx, dtype=np.int32)
    out['orbit_cam2world'] = np_(torch.cat(mats, 0))
    intr = torch.tensor([[[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]],
                         [[3.1, 0.2, 0.45], [0, 2.9, 0.55], [0, 0, 1]]])
    c2w = torch.cat([mats[1], mats[3]], 0)
    o, d = RaySampler()(c2w, intr, 5)
    out.update(rs_cam2world=np_(c2w), rs_intrinsics=np_(intr), rs_res=np.int32(5), rs_origins=np_(o), rs_dirs=np_(d))
    np.savez(os.path.join(HERE, 'camera.npz'), **out)
    print('camera.npz', sorted(out))


def make_ops_cases():
    from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu
    torch.manual_seed(5)
    out = {}
    # ---- bias_act: 9 activations x {plain, clamp} x grad order 0/1/2 (autograd of the ref path defines the grads)
    x0 = torch.randn(2, 5, 3, 4, dtype=torch.float64) * 2
    b0 = torch.randn(5, dtype=torch.float64)
    dy0 = torch.randn_like(x0)
    ddx0 = torch.randn_like(x0)
    out.update(ba_x=np_(x0), ba_b=np_(b0), ba_dy=np_(dy0), ba_ddx=np_(ddx0))
    for act in bias_act.activation_funcs:
        for clamp in (None, 0.9):
            x = x0.clone().requires_grad_(True)
            y = bias_act.bias_act(x, b0, dim=1, act=act, clamp=clamp, impl='ref')
            (dx,) = torch.autograd.grad(y, x, dy0, create_graph=True)
            if dx.requires_grad:
                (d2,) = torch.autograd.grad(dx, x, ddx0, allow_unused=True)
                d2 = torch.zeros_like(x0) if d2 is None else d2
            else:
                d2 = torch.zeros_like(x0)
            tag = f'ba_{act}_{"c" if clamp else "n"}'
            out.update({tag + '_y': np_(y), tag + '_dx': np_(dx), tag + '_d2': np_(d2)})
    xa = torch.randn(3, 7, dtype=torch.float32)
    out.update(ba2_x=np_(xa), ba2_b=np_(b0[:3].float()),
               ba2_y=np_(bias_act.bias_act(xa, b0[:3].float(), dim=0, act='lrelu', alpha=0.1, gain=0.7, impl='ref')))
    # ---- upfirdn2d
    x = torch.randn(2, 3, 9, 11, dtype=torch.float32)
    out['up_x'] = np_(x)
    f4 = upfirdn2d.setup_filter([1, 3, 3, 1])
    fa = torch.tensor([[1., 2., 0.5], [-1., 3., 0.25], [0.5, 0.1, -2.], [0.3, 0.2, 0.1]])      # asymmetric 4x3
    fs = upfirdn2d.setup_filter([1., 2., 3., 4., 3., 2., 1., 0.5])                                # separable (8 taps)
    out.update(up_f4=np_(f4), up_fa=np_(fa), up_fs=np_(fs))
    cases = {
        'blur':      dict(f=f4, up=1, down=1, padding=[1, 1, 1, 1], gain=4.0),
        'up2':       dict(f=f4, up=2, down=1, padding=[2, 1, 2, 1], gain=4.0),
        'down2':     dict(f=f4, up=1, down=2, padding=[1, 1, 1, 1], gain=1.0),
        'asym':      dict(f=fa, up=[2, 1], down=[1, 2], padding=[1, 2, 3, 0], gain=1.5),
        'asym_flip': dict(f=fa, up=[2, 1], down=[1, 2], padding=[1, 2, 3, 0], gain=1.5, flip_filter=True),
        'crop':      dict(f=f4, up=2, down=1, padding=[-1, 2, 3, -2], gain=1.0),
        'sep':       dict(f=fs, up=2, down=3, padding=[4, 3, 5, 2], gain=2.0),
        'sep_flip':  dict(f=fs, up=1, down=1, padding=[4, 3, 4, 3], gain=1.0, flip_filter=True),
        'none':      dict(f=None, up=2, down=1, padding=0, gain=1.0),
    }
    for k, kw in cases.items():
        out['up_' + k] = np_(upfirdn2d.upfirdn2d(x, impl='ref', **kw))
    out['up_filter2d'] = np_(upfirdn2d.filter2d(x, f4, impl='ref'))
    out['up_upsample2d'] = np_(upfirdn2d.upsample2d(x, f4, impl='ref'))
    out['up_downsample2d'] = np_(upfirdn2d.downsample2d(x, f4, impl='ref'))
    # gradient of the blur and up2 cases w.r.t. x (backward = same op with up<->down, flipped filter)
    for k in ('blur', 'up2', 'down2', 'asym'):
        xg = x.clone().requires_grad_(True)
        y = upfirdn2d.upfirdn2d(xg, impl='ref', **cases[k])
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(3))
        (dx,) = torch.autograd.grad(y, xg, g)
        out[f'up_{k}_dy'] = np_(g)
        out[f'up_{k}_dx'] = np_(dx)
    # ---- filtered_lrelu (ref path)
    xf = torch.randn(2, 3, 8, 8, dtype=torch.float32)
    bf = torch.randn(3, dtype=torch.float32)
    fu = upfirdn2d.setup_filter([1., 4., 6., 4., 1., 0.5]) * 0.9
    fu = fu if fu.ndim == 1 else fu[0] / fu[0].sum()
    fu12 = torch.tensor([0.02, 0.05, 0.1, 0.15, 0.2, 0.18, 0.12, 0.08, 0.05, 0.03, 0.01, 0.01])
    fd12 = fu12.flip(0) * 1.0
    out.update(fl_x=np_(xf), fl_b=np_(bf), fl_fu=np_(fu12), fl_fd=np_(fd12))
    out['fl_up2_down2'] = np_(filtered_lrelu.filtered_lrelu(xf, fu=fu12, fd=fd12, b=bf, up=2, down=2, padding=[10, 10, 10, 10],
                                                            gain=1.3, slope=0.1, clamp=0.8, impl='ref'))
    out['fl_up4_down2'] = np_(filtered_lrelu.filtered_lrelu(xf, fu=fu12, fd=fd12, b=bf, up=4, down=2, padding=[11, 10, 9, 12],
                                                            flip_filter=True, impl='ref'))
    out['fl_plain'] = np_(filtered_lrelu.filtered_lrelu(xf, b=bf, impl='ref'))
    np.savez(os.path.join(HERE, 'ops.npz'), **out)
    print('ops.npz', len(out), 'arrays')


def _ffhq_g_kwargs():
    """G_kwargs as train.py:238-377 assembles them from its option defaults (FFHQ configuration)."""
    rendering = {
        'image_resolution': 512, 'disparity_space_sampling': False, 'clamp_mode': 'softplus',
        'superresolution_module': 'training.superresolution.SuperresolutionHybrid8XDC',
        'c_gen_conditioning_zero': False, 'gpc_reg_prob': 0.5, 'c_scale': 1, 'superresolution_noise_mode': 'none',
        'density_reg': 0.25, 'density_reg_p_dist': 0.004, 'reg_type': 'l1', 'decoder_lr_mul': 1, 'sr_antialias': True,
        'depth_resolution': 48, 'depth_resolution_importance': 48,
        'ray_start': 2.25, 'ray_end': 3.3, 'box_warp': 1, 'avg_camera_radius': 2.7, 'avg_camera_pivot': [0, 0, 0.2],
    }
    return dict(z_dim=512, w_dim=512, c_dim=25, img_resolution=512, img_channels=3,
                mapping_kwargs=dict(num_layers=2), channel_base=32768, channel_max=512, fused_modconv_default='inference_only',
                rendering_kwargs=rendering, num_fp16_res=0, conv_clamp=None, sr_num_fp16_res=4,
                sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only', w_dim=512))


def make_generator_cases():
    """BASELINE configs 3 and 5 from the reference's own classes on the CPU (fp32; the reference forces fp32 for CPU tensors):
    generator_n4.npz  TriPlaneGenerator.synthesis at N=4, render resolution 64, noise_mode='const' (config 3);
    train_step.npz, train_step_grads.npz    one G + D step of training_loop.py:314-437 (SSIM / VGG terms dropped) on the same
                      generator in training mode and Discriminator(c_dim=25, img_resolution=64, img_channels=1,
                      mbstd_group_size=4): loss terms and the norm of every parameter's gradient.
    Weights, noise and the batch are functions of names / call indices (det_init.py), so only outputs are stored."""
    import det_init as DI
    from training.triplane import TriPlaneGenerator
    from training.networks_stylegan2 import Discriminator
    torch.set_num_threads(os.cpu_count() or 1)
    torch.manual_seed(0)
    G = DI.det_init_(TriPlaneGenerator(**_ffhq_g_kwargs()), 'G/').eval().requires_grad_(False)
    batch = DI.synthetic_batch(4)
    with torch.no_grad(), DI.DetNoise('config3'):
        ws = G.mapping(batch['z'], batch['c'])
        out = G.synthesis(ws, batch['c'], noise_mode='const', neural_rendering_resolution=64)
        planes_absmax = float(G.backbone.synthesis(ws, noise_mode='const').abs().max())
    img = out['image']
    np.savez_compressed(os.path.join(HERE, 'generator_n4.npz'),
                        ws_first=np_(ws[:, 0, :8]), image_raw=np_(out['image_raw']), image_depth=np_(out['image_depth']),
                        image_sub=np_(img[:, :, 4::8, 4::8]), image_mean=np_(img.mean((1, 2, 3))), image_std=np_(img.std((1, 2, 3))),
                        image_absmax=np.float32(img.abs().max()), planes_absmax=np.float32(planes_absmax))
    print('generator_n4: image std', float(img.std()), 'absmax', float(img.abs().max()), 'raw absmax', float(out['image_raw'].abs().max()))

    # ---- one training step (training_loop.py:314-437)
    import torch.nn.functional as F
    D = DI.det_init_(Discriminator(c_dim=25, img_resolution=64, img_channels=1, channel_base=32768, channel_max=512, num_fp16_res=4,
                                   conv_clamp=256, block_kwargs={}, mapping_kwargs={}, epilogue_kwargs=dict(mbstd_group_size=4)), 'D/')
    D.train().requires_grad_(False)
    G.train().requires_grad_(True)
    res, r1_gamma = 64, 1.0
    # The reference's block64 adds its ToRGB output IN PLACE into a view of the feature image that its first convolution
    # saved (superresolution.py:295 with triplane.py:86, networks_stylegan2.py:460): fine with fp16 blocks (the cast copies),
    # an autograd error in fp32.  allow_mutation_on_saved_tensors gives the gradient of the values that were actually used.
    with DI.DetNoise('config5'), torch.autograd.graph.allow_mutation_on_saved_tensors():
        ws = G.mapping(batch['z'], batch['c'], update_emas=False)                                                   # :333
        gen = G.synthesis(ws, batch['c'], neural_rendering_resolution=res, update_emas=False)                       # :335
        real = batch['loss_image']
        real_raw = F.interpolate(real, size=(res, res), mode='bilinear', align_corners=False, antialias=True)       # ssim_resize :180,:337
        l1 = (real - gen['image']).abs().mean((1, 2, 3))                                                            # :349
        l1_raw = (real_raw - gen['image_raw']).abs().mean((1, 2, 3))                                                # :343
        factor = batch['factor']
        gen_logits = D(gen['image_depth'], batch['c'])                                                              # :371
        loss_gan = F.softplus(-gen_logits).mean()
        loss = ((l1 + l1_raw) * factor).sum() / (factor.sum() + 1e-6) + 1.2 * loss_gan                              # :374
        loss.backward()
    g_names = [n for n, p_ in G.named_parameters() if p_.grad is not None]
    g_norms = np.array([float(p_.grad.double().norm()) for n, p_ in G.named_parameters() if p_.grad is not None])
    G.requires_grad_(False)
    D.requires_grad_(True)                                                                                          # :402-423
    d_gen_logits = D(gen['image_depth'].detach(), batch['c'])
    loss_dgen = F.softplus(d_gen_logits)
    loss_dgen.mean().backward()
    real_depth = batch['depth_image'].detach().requires_grad_(True)
    real_logits = D(real_depth, batch['condition_c'])
    loss_dreal = F.softplus(-real_logits)
    r1, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_depth], create_graph=True, only_inputs=True)
    loss_r1 = r1.square().sum([1, 2, 3]) * (r1_gamma / 2)
    (loss_dreal + loss_r1).mean().backward()
    d_names = [n for n, p_ in D.named_parameters()]
    d_norms = np.array([float(p_.grad.double().norm()) for n, p_ in D.named_parameters()])
    np.savez_compressed(os.path.join(HERE, 'train_step.npz'),
                        loss=np_(loss), l1=np_(l1), l1_raw=np_(l1_raw), loss_gan=np_(loss_gan), gen_logits=np_(gen_logits),
                        d_gen_logits=np_(d_gen_logits), real_logits=np_(real_logits), loss_dgen=np_(loss_dgen.mean()),
                        loss_dreal=np_(loss_dreal.mean()), loss_r1=np_(loss_r1), r1_gamma=np.float32(r1_gamma),
                        image_raw=np_(gen['image_raw']), image_depth=np_(gen['image_depth']), image_sub=np_(gen['image'][:, :, 4::8, 4::8]),
                        g_names=np.array(g_names), g_grad_norms=g_norms, d_names=np.array(d_names), d_grad_norms=d_norms,
                        g_grad_decoder_w1=np_(G.decoder.net[0].weight.grad), d_grad_out_w=np_(D.b4.out.weight.grad[:4, :64]))
    # three WHOLE gradient tensors for elementwise comparison (a sign or permutation error inside a tensor keeps its norm): one of the
    # backbone, one of the superresolution, one of the discriminator.  They are stored as float16 of value / max|value|
    # (0.3 / 0.6 / 0.5 MB instead of twice that; the 2^-11 storage rounding is a relative L2 error of 3e-4, inside the tests' 1e-3).
    gp = dict(G.named_parameters())
    big = {'g_backbone_b256_conv1_w': gp['backbone.synthesis.b256.conv1.weight'].grad, 'g_sr_block1_conv0_w': gp['superresolution.block1.conv0.weight'].grad,
           'd_b4_out_w': D.b4.out.weight.grad}
    packed = {}
    for k, t in big.items():
        scale = float(t.abs().max())
        packed[k + '_f16'] = (t / scale).to(torch.float16).numpy()
        packed[k + '_scale'] = np.float32(scale)
    np.savez_compressed(os.path.join(HERE, 'train_step_grads.npz'), **packed)
    print('train_step: loss', float(loss), 'gan', float(loss_gan), 'r1', float(loss_r1.mean()), '|dG|', float(np.sqrt((g_norms ** 2).sum())),
          '|dD|', float(np.sqrt((d_norms ** 2).sum())))


def main():
    _import_reference()
    sys.path.insert(0, HERE)
    if len(sys.argv) > 1 and sys.argv[1] == 'generator':          # the two slow ones alone
        return make_generator_cases()
    # density noise (renderer.py:146-147; rendering_kwargs carries no such key in train.py / gen_videos.py: the option exists upstream)
    make_render_case('render_dnoise.npz', seed=5, N=2, res=4, S=12, F=12, plane_hw=(12, 12), plane_scale=2.0, extra_opts=dict(density_noise=0.5))
    if len(sys.argv) > 1 and sys.argv[1] == 'dnoise':
        return
    make_render_case('render_s12.npz', seed=1, N=2, res=8, S=12, F=12, plane_hw=(16, 16), plane_scale=2.0)
    make_render_case('render_s48.npz', seed=2, N=1, res=4, S=48, F=48, plane_hw=(20, 24), plane_scale=1.0)
    make_render_case('render_misc.npz', seed=3, N=1, res=4, S=8, F=8, plane_hw=(8, 8), plane_scale=3.0,
                     extra_opts=dict(white_back=True, disparity_space_sampling=True), lr_mul=0.5)
    make_render_case('render_nofine.npz', seed=4, N=1, res=3, S=16, F=0, plane_hw=(8, 8), plane_scale=2.0)
    make_stage_cases()
    make_camera_cases()
    make_ops_cases()
    make_generator_cases()


if __name__ == '__main__':
    main()
