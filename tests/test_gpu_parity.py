"""GPU parity tests (run with -m gpu on an MI355X): the hand-written HIP path, called through the C ABI
(ctypes -> libgnerf_hip.so), against the CPU oracle, the golden vectors captured from the reference, and
size-independent properties at BASELINE.json's full sizes.

Tolerances (fp32 arithmetic on both sides, different summation orders and hardware exp2/log2 in the MLP
activations): rgb pixel MSE < 1e-8 against the oracle here -- north_star's bound is 1e-4."""

import ctypes
import math

import numpy as np
import pytest
import torch

from conftest import has_gpu

pytestmark = pytest.mark.gpu

RENDER_CASES = ['render_s12.npz', 'render_s48.npz', 'render_misc.npz', 'render_nofine.npz', 'render_dnoise.npz']       # (the last: density noise, renderer.py:146-147)


@pytest.fixture(scope='module')
def dev():
    if not has_gpu():
        pytest.fail('GPU tests selected but no GPU is visible (the HIP path has no CPU fallback)')
    import gnerf_hip
    gnerf_hip.load()
    return torch.device('cuda', 0)


def _t(a, dev=None, dt=torch.float32):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    return t if dev is None else t.to(dev)


def _fold(g):
    from oracle import render_ref as R
    return R.fold_decoder(_t(g['w1']), _t(g['b1']), _t(g['w2']), _t(g['b2']), float(g['lr_mul']))


def _decoder_module(g, dev):
    """gnerf_harness.TriPlaneDecoder (the OSGDecoder of triplane.py:113-136) holding a fixture's raw weights."""
    import gnerf_harness as H
    dec = H.TriPlaneDecoder(decoder_lr_mul=float(g['lr_mul'])).to(dev).requires_grad_(False)
    with torch.no_grad():
        dec.net[0].weight.copy_(_t(g['w1'], dev)); dec.net[0].bias.copy_(_t(g['b1'], dev))
        dec.net[2].weight.copy_(_t(g['w2'], dev)); dec.net[2].bias.copy_(_t(g['b2'], dev))
    return dec


def _hip_render(g, dev, debug=False, image_width=None):
    import gnerf_hip
    dec = [t.to(dev) for t in _fold(g)]
    N = g['planes'].shape[0]
    nhwc = gnerf_hip.planes_to_nhwc(_t(g['planes'], dev))
    nf = _t(g['noise_fine'], dev) if 'noise_fine' in g else None
    res = int(g['res'])
    return gnerf_hip.render_forward(
        nhwc, N, dec, _t(g['ray_origins'], dev), _t(g['ray_dirs'], dev), _t(g['noise_coarse'], dev), nf,
        depth_resolution=int(g['depth_resolution']), depth_resolution_importance=int(g['depth_resolution_importance']),
        ray_start=float(g['ray_start']), ray_end=float(g['ray_end']), box_warp=float(g['box_warp']),
        white_back=bool(g['white_back']), disparity_space_sampling=bool(g['disparity']),
        image_width=res if image_width is None else image_width, debug=debug,
        sigma_noise=(_t(g['sigma_noise_coarse'], dev), _t(g['sigma_noise_fine'], dev)) if 'sigma_noise_coarse' in g else None)


# ---------------------------------------------------------------------------- layout + rays


def test_planes_to_nhwc(dev):
    import gnerf_hip
    for shape in [(2, 3, 32, 16, 16), (1, 3, 32, 20, 24), (1, 3, 32, 7, 5), (6, 40, 9, 13)]:
        x = torch.randn(*shape, device=dev)
        out = gnerf_hip.planes_to_nhwc(x)
        ref = x.reshape(-1, *shape[-3:]).permute(0, 2, 3, 1).contiguous()
        assert torch.equal(out, ref)


def test_planes_from_nhwc_roundtrip(dev):
    import gnerf_hip
    for shape in [(2, 3, 32, 16, 16), (1, 3, 32, 20, 24), (1, 3, 32, 7, 5), (6, 40, 9, 13)]:
        x = torch.randn(*shape, device=dev)
        nhwc = gnerf_hip.planes_to_nhwc(x)
        back = gnerf_hip.planes_from_nhwc(nhwc)
        assert torch.equal(back, x.reshape(-1, *shape[-3:]))
    assert gnerf_hip.planes_from_nhwc(gnerf_hip.planes_to_nhwc(torch.randn(2, 3, 32, 8, 8, device=dev)), 2).shape == (2, 3, 32, 8, 8)


def test_make_rays(dev, golden):
    import gnerf_hip
    g = golden('camera.npz')
    o, d = gnerf_hip.make_rays(_t(g['rs_cam2world'], dev), _t(g['rs_intrinsics'], dev), int(g['rs_res']))
    np.testing.assert_allclose(o.cpu().numpy(), g['rs_origins'], atol=0)
    np.testing.assert_allclose(d.cpu().numpy(), g['rs_dirs'], atol=2.4e-7)
    from training.volumetric_rendering.ray_sampler import RaySampler
    o2, d2 = RaySampler()(_t(g['rs_cam2world'], dev), _t(g['rs_intrinsics'], dev), int(g['rs_res']))
    assert torch.equal(o, o2) and torch.equal(d, d2)


# ---------------------------------------------------------------------------- renderer


@pytest.mark.parametrize('case', RENDER_CASES)
def test_render_golden_stage_by_stage(dev, golden, case):
    import gnerf_hip
    g = golden(case)
    rgb, depth, wsum, dbg = _hip_render(g, dev, debug=True)
    torch.cuda.synchronize()
    dbg = dbg.cpu().numpy()
    N, M = g['out_rgb'].shape[:2]
    S, F = int(g['depth_resolution']), int(g['depth_resolution_importance'])
    dbg = dbg.reshape(N, M, gnerf_hip.DEBUG_SLOTS, S + F)
    np.testing.assert_allclose(dbg[:, :, 0, :S], g['depths_coarse'], rtol=0, atol=5e-7)      # <= 2 ulp at depth ~3.5 (the disparity form divides)
    np.testing.assert_allclose(dbg[:, :, 1, :S], g['sigma_coarse'], rtol=1e-4, atol=5e-5)
    if F > 0:
        np.testing.assert_allclose(dbg[:, :, 2, :S - 1], g['weights_coarse'], rtol=1e-3, atol=2e-6)
        np.testing.assert_allclose(dbg[:, :, 3, :F], g['depths_fine'], rtol=0, atol=5e-5)
        np.testing.assert_allclose(dbg[:, :, 4, :F], g['sigma_fine'], rtol=1e-3, atol=3e-4)
        np.testing.assert_allclose(dbg[:, :, 5, :], g['depths_all'], rtol=0, atol=5e-5)
    np.testing.assert_allclose(rgb.cpu().numpy(), g['out_rgb'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), g['out_depth'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(wsum.cpu().numpy(), g['out_wsum'], rtol=0, atol=1e-4)
    assert float(((rgb.cpu().numpy() - g['out_rgb']) ** 2).mean()) < 1e-8


@pytest.mark.parametrize('kernel', ['pipe', 'coop', 'generic'])
def test_render_density_noise_in_every_kernel(dev, golden, monkeypatch, kernel):
    """renderer.py:146-147 (`sigma += randn_like(sigma) * density_noise` in run_model) inside the fused kernels (round 6: it used to send a
    GPU call to the PyTorch-op form): the recorded draws of the reference-made fixture through each forward kernel, against the fixture,
    and the drop-in class -- which must consume the generator in the reference's order: rand_like, randn_like, rand, randn_like."""
    import gnerf_hip
    g = golden('render_dnoise.npz')
    monkeypatch.setenv('GNERF_RENDER_KERNEL', kernel)
    rgb, depth, wsum = _hip_render(g, dev)
    np.testing.assert_allclose(rgb.cpu().numpy(), g['out_rgb'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), g['out_depth'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(wsum.cpu().numpy(), g['out_wsum'], rtol=0, atol=1e-4)
    assert float(((rgb.cpu().numpy() - g['out_rgb']) ** 2).mean()) < 1e-8
    # without the noise the result is another one (the option is not silently dropped)
    g0 = {k: v for k, v in g.items() if not k.startswith('sigma_noise')}
    assert float((_hip_render(g0, dev)[0] - rgb).abs().max()) > 1e-3
    if kernel != 'pipe':
        return
    # the class: no fallback warning, the reference's draw order (replayed from the recorded draws through the global generator's hooks)
    import warnings
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from oracle import render_ref as R
    dec = _decoder_module(g, dev)
    draws = [_t(g['noise_coarse'], dev).reshape(g['noise_coarse'].shape + (1,)), _t(g['sigma_noise_coarse'], dev) / float(g['density_noise']),
             _t(g['noise_fine'], dev), _t(g['sigma_noise_fine'], dev) / float(g['density_noise'])]
    kinds = []
    real_rand, real_randn = torch.rand, torch.randn

    def fake_rand(*a, **k):
        kinds.append('rand')
        return draws[len(kinds) - 1].reshape(*(a[0] if isinstance(a[0], (list, tuple)) else a))
    def fake_randn(*a, **k):
        kinds.append('randn')
        return draws[len(kinds) - 1].reshape(*(a[0] if isinstance(a[0], (list, tuple)) else a))
    monkeypatch.setattr(torch, 'rand', fake_rand)
    monkeypatch.setattr(torch, 'randn', fake_randn)
    opts = dict(depth_resolution=int(g['depth_resolution']), depth_resolution_importance=int(g['depth_resolution_importance']), ray_start=float(g['ray_start']),
                ray_end=float(g['ray_end']), box_warp=float(g['box_warp']), clamp_mode='softplus', disparity_space_sampling=False, white_back=False,
                density_noise=float(g['density_noise']))
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter('error')
        out = ImportanceRenderer()(_t(g['planes'], dev), dec, _t(g['ray_origins'], dev), _t(g['ray_dirs'], dev), opts)
    monkeypatch.setattr(torch, 'rand', real_rand)
    monkeypatch.setattr(torch, 'randn', real_randn)
    assert kinds == ['rand', 'randn', 'rand', 'randn'], kinds
    np.testing.assert_allclose(out[0].cpu().numpy(), g['out_rgb'], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out[1].cpu().numpy(), g['out_depth'], rtol=0, atol=1e-4)


@pytest.mark.parametrize('case', RENDER_CASES[:2])
def test_render_ray_order_independent(dev, golden, case):
    """The image-tile walk (image_width hint) and the linear walk must give bit-identical rays."""
    g = golden(case)
    a = _hip_render(g, dev)
    b = _hip_render(g, dev, image_width=0)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def _random_scene(seed, N, res, S, F, hw, scale=1.5):
    from oracle import render_ref as R
    gen = torch.Generator().manual_seed(seed)
    planes = torch.randn(N, 3, 32, hw[0], hw[1], generator=gen) * scale
    dec = R.fold_decoder(torch.randn(64, 32, generator=gen), torch.randn(64, generator=gen) * 0.2,
                         torch.randn(33, 64, generator=gen), torch.randn(33, generator=gen) * 0.2)
    c2w = torch.cat([R.lookat_pose(3.14 / 2 + 0.5 * np.sin(1.0 + i), 3.14 / 2 - 0.05 + 0.2 * np.cos(2.0 * i), 2.7) for i in range(N)])
    intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1)
    o, d = R.make_rays(c2w, intr, res)
    nc = torch.rand(N, res * res, S, generator=gen)
    nf = torch.rand(N * res * res, max(F, 1), generator=gen)[:, :F]
    return planes, dec, o, d, nc, nf


@pytest.mark.parametrize('cfg', [
    dict(N=2, res=16, S=48, F=48, hw=(64, 64)),             # the headline sampling config at a size the oracle does in seconds
    dict(N=1, res=12, S=96, F=96, hw=(32, 48)),             # gen_videos.py doubles the sample counts (gen_videos.py:127-128)
    dict(N=1, res=5, S=17, F=30, hw=(9, 11)),               # ragged everything: odd res (linear ray walk), partial MLP tiles
    dict(N=3, res=4, S=64, F=3, hw=(16, 16)),
    dict(N=1, res=4, S=4, F=5, hw=(4, 4)),                  # smallest importance-sampled case (S-3 = 1 pdf bin)
    dict(N=1, res=4, S=2, F=0, hw=(4, 4)),                  # smallest case at all
    dict(N=1, res=4, S=130, F=100, hw=(8, 8)),              # beyond 96+96: three-tiles-per-wave pipelined kernel, ragged tiles
    dict(N=1, res=8, S=128, F=128, hw=(16, 16)),            # gen_videos.py's doubling of the ShapeNet config's 64+64
    dict(N=1, res=4, S=144, F=144, hw=(8, 8)),              # ... its limit
    dict(N=1, res=4, S=97, F=1, hw=(8, 8)),                 # ... seven coarse tiles, one fine sample
    dict(N=1, res=4, S=145, F=20, hw=(8, 8)),               # one sample past it: one-wave-per-ray kernel
    dict(N=1, res=2, S=256, F=256, hw=(8, 8)),              # GNERF_MAX_SAMPLES on both passes
    dict(N=2, res=8, S=64, F=64, hw=(16, 12)),              # ShapeNet config's sample counts (train.py:353-354)
    dict(N=1, res=6, S=50, F=70, hw=(16, 16)),              # two-tiles-per-wave pipelined kernel, ragged on both passes
    dict(N=2, res=4, S=96, F=5, hw=(8, 8)),                 # ... six coarse tiles, a sliver of a fine tile
    dict(N=1, res=5, S=33, F=96, hw=(12, 8)),               # ... three coarse tiles (one nearly empty), six fine
])
def test_render_vs_oracle(dev, cfg):
    import gnerf_hip
    from oracle import render_ref as R
    planes, dec, o, d, nc, nf = _random_scene(7, **cfg)
    S, F, N = cfg['S'], cfg['F'], cfg['N']
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    rgb, depth, wsum = gnerf_hip.render_forward(nhwc, N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None,
                                                depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                                                image_width=cfg['res'])
    mse = float(((rgb.cpu() - ref_rgb) ** 2).mean())
    assert mse < 1e-8, mse
    np.testing.assert_allclose(rgb.cpu().numpy(), ref_rgb.numpy(), atol=2e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), ref_depth.numpy(), atol=2e-4)
    np.testing.assert_allclose(wsum.cpu().numpy(), ref_w.numpy(), atol=2e-4)


@pytest.mark.parametrize('S,F', [(48, 48), (24, 40), (96, 96), (128, 128)])
def test_render_tied_fine_depths(dev, S, F):
    """Equal uniform draws give bit-identical fine depths: the merge must still produce a permutation (stable order,
    like torch.sort on the concatenation).  Exercises the tie-repair path of the merge in every kernel."""
    import gnerf_hip
    from oracle import render_ref as R
    planes, dec, o, d, nc, nf = _random_scene(11, N=1, res=8, S=S, F=F, hw=(16, 16))
    nf = nf.clone()
    nf[:, 5] = nf[:, 9]
    nf[:, 20] = nf[:, 21]
    nf[:, 30] = nf[:, 21]
    nf[3, :] = nf[3, 0]                       # one ray whose fine samples all coincide
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    rgb, depth, wsum = gnerf_hip.render_forward(nhwc, 1, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev),
                                                depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=8)
    assert float(((rgb.cpu() - ref_rgb) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(depth.cpu().numpy(), ref_depth.numpy(), atol=2e-4)
    np.testing.assert_allclose(wsum.cpu().numpy(), ref_w.numpy(), atol=2e-4)


@pytest.mark.parametrize('kernel,S', [('pipe', 48), ('pipe', 96), ('coop', 48), ('generic', 48)])
def test_render_coarse_depths_swapped_by_rounding(dev, kernel, S, monkeypatch):
    """Jitter u = 1 - 2^-24 makes lin_k + u*delta round one ulp past the next proposal (7 of the 47 neighbour pairs at
    the default limits): the coarse depths are then NOT ascending, and the merge must still be the reference's stable
    sort -- sorted depths non-decreasing, every slot written."""
    import gnerf_hip
    from oracle import render_ref as R
    monkeypatch.setenv('GNERF_RENDER_KERNEL', kernel)
    N, res, F = 1, 8, S
    planes, dec, o, d, nc, nf = _random_scene(21, N, res, S, F, (32, 32))
    nc = nc.clone()
    nc[:, 0::2, 0::2] = 1.0 - 2.0 ** -24
    nc[:, 1::2, 1::2] = 1.0 - 2.0 ** -24
    nc[:, 1::2, 0::2] = 0.0
    nc[:, 0::2, 1::2] = 0.0
    rs, re = (2.25, 3.3) if S == 48 else (2.251, 3.3007)          # limits at which the rounding swaps neighbours for this S
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=rs, ray_end=re, box_warp=1.0, clamp_mode='softplus')
    st = {}
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf, stages=st)
    assert bool((st['depths_coarse'][:, 1:] < st['depths_coarse'][:, :-1]).any())          # the premise: swapped neighbours exist
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    rgb, depth, wsum, dbg = gnerf_hip.render_forward(nhwc, N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev),
                                                     depth_resolution=S, depth_resolution_importance=F, ray_start=rs, ray_end=re, box_warp=1.0,
                                                     image_width=res, debug=True)
    sorted_d = dbg[:, gnerf_hip.DBG_DEPTH_SORTED if hasattr(gnerf_hip, 'DBG_DEPTH_SORTED') else 5].cpu()
    assert bool((sorted_d[:, 1:] >= sorted_d[:, :-1]).all())
    np.testing.assert_allclose(sorted_d.numpy(), st['depths_all'].numpy(), atol=5e-7)
    np.testing.assert_allclose(rgb.cpu().numpy(), ref_rgb.numpy(), atol=2e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), ref_depth.numpy(), atol=2e-4)


def test_render_per_ray_limits_vs_oracle(dev):
    """'auto' ray limits: per-ray start/end tensors (renderer.py:93-98, math_utils.linspace)."""
    import gnerf_hip
    from oracle import render_ref as R
    planes, dec, o, d, nc, nf = _random_scene(9, N=2, res=8, S=24, F=24, hw=(16, 16))
    gen = torch.Generator().manual_seed(1)
    rs = 2.0 + 0.4 * torch.rand(2 * 64, generator=gen)
    re = rs + 0.5 + torch.rand(2 * 64, generator=gen)
    opts = dict(depth_resolution=24, depth_resolution_importance=24, ray_start=rs, ray_end=re, box_warp=1.0, clamp_mode='softplus')
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    rgb, depth, wsum = gnerf_hip.render_forward(nhwc, 2, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev),
                                                depth_resolution=24, depth_resolution_importance=24, ray_start=rs.to(dev), ray_end=re.to(dev),
                                                box_warp=1.0, image_width=8)
    assert float(((rgb.cpu() - ref_rgb) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(depth.cpu().numpy(), ref_depth.numpy(), atol=2e-4)


def test_render_zero_weight_ray_takes_global_max_depth(dev):
    """ray_marcher.py:49-50: a ray whose weights underflow to 0 gets depth = max over ALL depths of the call."""
    import gnerf_hip
    from oracle import render_ref as R
    planes, dec, o, d, nc, nf = _random_scene(3, N=1, res=4, S=12, F=12, hw=(8, 8))
    w1, b1, w2, b2 = [t.clone() for t in dec]
    w2[0] = 0
    b2[0] = -300.0              # density -300 everywhere: softplus(-301) underflows, every weight is exactly 0
    dec = (w1, b1, w2, b2)
    opts = dict(depth_resolution=12, depth_resolution_importance=12, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
    st = {}
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf, st)
    assert float(ref_w.abs().max()) == 0.0
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    rgb, depth, wsum = gnerf_hip.render_forward(nhwc, 1, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev),
                                                depth_resolution=12, depth_resolution_importance=12, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=4)
    assert float(wsum.abs().max()) == 0.0
    np.testing.assert_allclose(depth.cpu().numpy(), ref_depth.numpy(), atol=1e-5)
    assert torch.all(depth == depth.max())
    np.testing.assert_allclose(rgb.cpu().numpy(), ref_rgb.numpy(), atol=1e-6)


@pytest.mark.production_path
def test_render_full_size_properties(dev):
    """BASELINE.json config 2 (N=4, 128x128 rays, 48+48 samples, 256x256 planes): too big for the oracle in a
    unit test, so check properties: determinism, bounds, item independence (a batch renders each item
    exactly as a batch of one does), and a strided subset of rays against the oracle."""
    import gnerf_hip
    from oracle import render_ref as R
    N, res, S, F = 4, 128, 48, 48
    planes, dec, o, d, nc, nf = _random_scene(0, N=N, res=res, S=S, F=F, hw=(256, 256), scale=1.0)
    pl, de = planes.to(dev), [t.to(dev) for t in dec]
    od, dd, ncd, nfd = o.to(dev), d.to(dev), nc.to(dev), nf.to(dev)
    kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0)
    nhwc = gnerf_hip.planes_to_nhwc(pl)
    a = gnerf_hip.render_forward(nhwc, N, de, od, dd, ncd, nfd, image_width=res, **kw)
    for _ in range(6):                                       # bit-identical from run to run: no float atomics on the data path, and
        b = gnerf_hip.render_forward(nhwc, N, de, od, dd, ncd, nfd, image_width=res, **kw)       # no timing-dependent hazards (a missed
        for x, y in zip(a, b):                               # MFMA operand hazard once showed up here as 2 rays in 65536)
            assert torch.equal(x, y)
    rgb, depth, wsum = a
    assert torch.isfinite(rgb).all() and torch.isfinite(depth).all() and torch.isfinite(wsum).all()
    assert float(rgb.min()) >= -1.0021 and float(rgb.max()) <= 1.0021       # sigmoid*1.002-0.001 composited with weights <= 1
    assert float(wsum.min()) >= 0 and float(wsum.max()) <= 1.0 + 1e-5
    assert float(depth.min()) >= 2.25 and float(depth.max()) <= 3.3 + (3.3 - 2.25) / 47 + 1e-5
    # item independence: item 2 alone
    one = gnerf_hip.render_forward(nhwc[6:9].contiguous(), 1, de, od[2:3], dd[2:3], ncd[2:3], nfd[2 * res * res:3 * res * res], image_width=res, **kw)
    assert torch.equal(one[0][0], rgb[2]) and torch.equal(one[2][0], wsum[2])
    # a strided subset of rays against the oracle (same planes, those rays only)
    idx = torch.arange(0, res * res, 509)
    sub_nf = nf.reshape(N, res * res, F)[:, idx].reshape(-1, F)
    opts = dict(kw, clamp_mode='softplus')
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o[:, idx], d[:, idx], opts, nc[:, idx], sub_nf)
    assert float(((rgb.cpu()[:, idx] - ref_rgb) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(wsum.cpu()[:, idx].numpy(), ref_w.numpy(), atol=2e-4)
    # bench.py's producer-layout step at its own size: the interleaved [4,256,256,96] planes (channels_last memory of the
    # backbone's [N,96,H,W] output) with a caller-supplied max |planes|, and the NCHW-input step (repack + absmax inside the
    # step), must both give the [12,256,256,32] result bit for bit -- and therefore pass the oracle check above
    inter = pl.reshape(N, 96, 256, 256).permute(0, 2, 3, 1).contiguous()
    assert inter.shape == (4, 256, 256, 96)
    amax = gnerf_hip.planes_absmax(inter)
    assert float(amax) == float(pl.abs().max())
    c = gnerf_hip.render_forward(inter, N, de, od, dd, ncd, nfd, image_width=res, planes_absmax=amax, **kw)
    nhwc2, amax2 = gnerf_hip.planes_to_nhwc(pl, with_absmax=True)
    e = gnerf_hip.render_forward(nhwc2, N, de, od, dd, ncd, nfd, image_width=res, planes_absmax=amax2, **kw)
    assert float(amax2) == float(amax) and gnerf_hip.last_mlp_choice(dev) == 'f16x3'
    for x, y, z in zip(a, c, e):
        assert torch.equal(x, y) and torch.equal(x, z)


@pytest.mark.production_path
def test_render_full_size_on_backbone_planes(dev):
    """Config 2 on planes a generator produces (SURVEY 8d's "second run with real backbone output"; bench.py realistic_planes_step):
    the random-init FFHQ backbone's [4,96,256,256] channels_last output with the generator's default-init decoder.  Properties at
    full size -- the three plane routes (producer layout read in place with the producer's max |planes|; the same memory with the
    launcher measuring it; NCHW copy repacked to [12,256,256,32]) give the same bits, runs repeat bit for bit, the in-kernel
    rays / draws form agrees -- plus a strided subset of rays against the oracle on the same planes and decoder."""
    import gnerf_hip
    import gen_videos_mi355x as GV
    from oracle import render_ref as R
    from training.volumetric_rendering import renderer as RM
    N, res, S = 4, 128, 48
    with torch.no_grad():
        G = GV.build_random_generator(0, dev)
        z = torch.randn(N, G.z_dim, generator=torch.Generator().manual_seed(11)).to(dev)
        img = G.backbone.synthesis(G.mapping(z, torch.zeros(N, 25, device=dev)), noise_mode='const')
        planes5 = img.view(N, 3, 32, 256, 256)
        inter, amax = G.renderer._planes_nhwc(planes5)
        assert inter.shape == (N, 256, 256, 96) and inter.data_ptr() == img.data_ptr()         # the producer's memory, read in place
        dec = G.renderer._decoder_cache(RM._osg_decoder_weights(G.decoder))
    c2w = torch.cat([R.lookat_pose(3.14 / 2 + 0.3 * i, 3.14 / 2 - 0.05 * i, 2.7) for i in range(N)]).to(dev)
    intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1).to(dev)
    o, d = gnerf_hip.make_rays(c2w, intr, res)
    kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)
    torch.manual_seed(3)
    nc, nf = torch.rand([N, res * res, S, 1], device=dev), torch.rand(N * res * res, S, device=dev)
    a = gnerf_hip.render_forward(inter, N, dec, o, d, nc, nf, planes_absmax=amax, **kw)
    choice = gnerf_hip.last_mlp_choice(dev)
    assert choice in ('f16x3', 'f32')
    assert float(amax) == float(planes5.abs().max())
    b = gnerf_hip.render_forward(inter, N, dec, o, d, nc, nf, **kw)                              # the launcher measures max |planes|
    nchw = planes5.contiguous()                                                                  # a plain NCHW copy -> repack inside
    nhwc2, amax2 = gnerf_hip.planes_to_nhwc(nchw, with_absmax=True)
    c = gnerf_hip.render_forward(nhwc2, N, dec, o, d, nc, nf, planes_absmax=amax2, **kw)
    e = gnerf_hip.render_forward(inter, N, dec, o, d, nc, nf, planes_absmax=amax, **kw)
    for x, y, z_, w in zip(a, b, c, e):
        assert torch.equal(x, y) and torch.equal(x, z_) and torch.equal(x, w)
    assert gnerf_hip.last_mlp_choice(dev) == choice
    # rays and draws made in the kernel: same generator state -> same bits
    torch.manual_seed(3)
    plan = gnerf_hip.torch_philox_plan(dev, N, res * res, S, S)
    g = gnerf_hip.render_forward(inter, N, dec, None, None, None, None, planes_absmax=amax, cameras=(c2w, intr, res), rng=plan, **kw)
    for x, y in zip(a, g):
        assert torch.equal(x, y)
    rgb, depth, wsum = a
    assert torch.isfinite(rgb).all() and torch.isfinite(depth).all() and float(wsum.min()) >= 0 and float(wsum.max()) <= 1.0 + 1e-5
    idx = torch.arange(0, res * res, 701)
    ncc, nfc = nc.cpu().reshape(N, res * res, S), nf.cpu().reshape(N, res * res, S)
    ref_rgb, ref_depth, ref_w = R.render(nchw.cpu(), [t.cpu() for t in dec], o.cpu()[:, idx], d.cpu()[:, idx], dict(kw, clamp_mode='softplus'),
                                         ncc[:, idx], nfc[:, idx].reshape(-1, S))
    assert float(((rgb.cpu()[:, idx] - ref_rgb) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(wsum.cpu()[:, idx].numpy(), ref_w.numpy(), atol=2e-4)


def test_render_fuzz_slice(dev):
    """The first 200 cases of tests/parity_tools/fuzz_render.py's seed-31 sequence (the sweep that profiles/r0N_fuzz.jsonl records at full
    length) inside the suite, at the suite's own criterion: rgb MSE < max(1e-8, 4 x the fp32 noise floor), every kernel, both plane
    layouts, a third of the cases at wild magnitudes (both decoder arithmetics)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'parity_tools'))
    import fuzz_render
    out = fuzz_render.run(200, 31, mult=4.0)
    assert out['failures'] == [], out['failures'][:3]
    assert out['worst']['mse'] < 1e-5 and out['worst']['wsum'] < 5e-3


def _scaled_scene(plane_scale, weight_scale, S, F, res=8):
    planes, dec, o, d, nc, nf = _random_scene(31, N=2, res=res, S=S, F=F, hw=(24, 20), scale=1.0)
    return planes * plane_scale, [t * weight_scale for t in dec], o, d, nc, nf


@pytest.mark.parametrize('S,F', [(48, 48), (96, 96), (40, 0)])                   # pipe<1>, pipe<2>, coop
@pytest.mark.parametrize('weight_scale', [1e-3, 1.0, 1e3])
@pytest.mark.parametrize('plane_scale', [1e-5, 1e-3, 1.0, 1e3, 1e5])
def test_render_decoder_arithmetic_is_range_safe(dev, plane_scale, weight_scale, S, F):
    """The f16 hi/lo decoder arithmetic only holds inside f16's range (|x| < 65520 or hi = inf, lo = NaN; below 2^-14 the
    compensation is lost).  The reference's decoder is fp32 (triplane.py:124-136, networks_stylegan2.py:121-134) and is
    finite for all of these inputs, so the default (mlp='auto') must be too, at the usual tolerance: the device-side choice
    sends out-of-range or ill-conditioned calls to the fp32-MFMA kernels."""
    import gnerf_hip
    from oracle import render_ref as R
    planes, dec, o, d, nc, nf = _scaled_scene(plane_scale, weight_scale, S, F)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf)
    assert torch.isfinite(ref_rgb).all() and torch.isfinite(ref_w).all()                    # the premise: fp32 is fine
    # How well is fp32 itself defined here?  At |planes| x |weights| ~ 1e5 the decoder's outputs are ~1e5 with ~0.03 of
    # rounding noise, and the handful of samples whose pre-sigmoid value lands within a few units of 0 differ between ANY two
    # fp32 summation orders (ATen's addmm on the CPU, MFMA here, cuBLAS upstream).  The oracle in float64 measures that floor:
    # the HIP path must be as close to the fp32 reference as the fp32 reference is to exact arithmetic (and < 1e-8 wherever
    # fp32 is well conditioned, which is every case but the largest magnitudes).
    ex_rgb, _, ex_w = R.render(planes.double(), [t.double() for t in dec], o.double(), d.double(), opts, nc.double(), nf.double())
    floor = float(((ref_rgb.double() - ex_rgb) ** 2).mean())
    w_floor = float((ref_w.double() - ex_w).abs().max())
    nhwc, amax = gnerf_hip.planes_to_nhwc(planes.to(dev), with_absmax=True)
    assert float(amax) == float(planes.abs().max())
    args = (nhwc, 2, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None)
    kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=8)
    for given in (amax, None):                                   # absmax handed over / measured by the call itself
        rgb, depth, wsum = gnerf_hip.render_forward(*args, planes_absmax=given, **kw)
        choice = gnerf_hip.last_mlp_choice(dev)
        assert torch.isfinite(rgb).all() and torch.isfinite(wsum).all() and torch.isfinite(depth).all(), choice
        mse = float(((rgb.cpu() - ref_rgb) ** 2).mean())
        assert mse < max(1e-8, 4 * floor), (mse, floor, choice)
        np.testing.assert_allclose(wsum.cpu().numpy(), ref_w.numpy(), atol=max(2e-4, 4 * w_floor))
        if floor < 1e-9:
            fin = torch.isfinite(ref_depth)
            np.testing.assert_allclose(depth.cpu()[fin].numpy(), ref_depth[fin].numpy(), atol=5e-4)
    # where the choice is forced by the bounds
    if plane_scale == 1.0 and weight_scale == 1.0:
        assert choice == 'f16x3'
    if plane_scale >= 1e5 or weight_scale >= 1e3:
        assert choice == 'f32'


@pytest.mark.parametrize('S,F', [(48, 48), (96, 96), (128, 128), (40, 0)])
def test_render_decoder_arithmetic_forced(dev, S, F):
    """Both shipped arithmetics agree with the oracle on an in-range scene when forced, and the forced f16 path shows the
    hazard the default guards against (non-finite output at |planes| ~ 1e5)."""
    import gnerf_hip
    from oracle import render_ref as R
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
    kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=8)
    planes, dec, o, d, nc, nf = _scaled_scene(1.0, 1.0, S, F)
    ref_rgb, _, ref_w = R.render(planes, dec, o, d, opts, nc, nf)
    args = (2, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    outs = {}
    for mlp in ('f16x3', 'f32'):
        rgb, depth, wsum = outs[mlp] = gnerf_hip.render_forward(nhwc, *args, mlp=mlp, **kw)
        assert float(((rgb.cpu() - ref_rgb) ** 2).mean()) < 1e-8
        np.testing.assert_allclose(wsum.cpu().numpy(), ref_w.numpy(), atol=2e-4)
    assert float((outs['f16x3'][0] - outs['f32'][0]).abs().max()) < 1e-4
    big = gnerf_hip.planes_to_nhwc((planes * 1e5).to(dev))
    bad = gnerf_hip.render_forward(big, *args, mlp='f16x3', **kw)[0]
    good = gnerf_hip.render_forward(big, *args, mlp='f32', **kw)[0]
    assert torch.isfinite(good).all() and not torch.isfinite(bad).all()
    with pytest.raises(RuntimeError):
        gnerf_hip.render_forward(nhwc, *args, mlp='bf16', **kw)


def test_planes_absmax(dev):
    import gnerf_hip
    for n in (1, 3, 4, 1000, 4097, 1 << 20):
        x = torch.randn(n, device=dev) * 3
        assert float(gnerf_hip.planes_absmax(x)) == float(x.abs().max())
    x = torch.randn(2, 3, 32, 9, 7, device=dev)
    x[1, 2, 5, 3, 3] = -77.5
    assert float(gnerf_hip.planes_to_nhwc(x, with_absmax=True)[1]) == 77.5
    x[0, 0, 0, 0, 0] = float('nan')
    assert torch.isnan(gnerf_hip.planes_to_nhwc(x, with_absmax=True)[1]).all() and torch.isnan(gnerf_hip.planes_absmax(x)).all()
    x[0, 0, 0, 0, 0] = float('inf')
    assert float(gnerf_hip.planes_absmax(x)) == float('inf')


def test_planes_absmax_contract_debug_check(dev, monkeypatch):
    """include/gnerf_hip.h: a caller-supplied planes_absmax must bound max |planes| of the call's planes.  GNERF_VERIFY_ABSMAX=1 makes
    gnerf_render_forward check it: an exact value and an upper bound pass, a stale (too small) value is refused."""
    import gnerf_hip
    planes, dec, o, d, nc, nf = _random_scene(3, N=1, res=4, S=48, F=48, hw=(16, 16))
    nhwc, amax = gnerf_hip.planes_to_nhwc(planes.to(dev), with_absmax=True)
    args = (nhwc, 1, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev))
    kw = dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=4)
    monkeypatch.setenv('GNERF_VERIFY_ABSMAX', '1')
    a = gnerf_hip.render_forward(*args, planes_absmax=amax, **kw)
    b = gnerf_hip.render_forward(*args, planes_absmax=amax * 1.0001, **kw)
    assert torch.equal(a[0], b[0])
    c = gnerf_hip.render_forward(*args, planes_absmax=amax * 3, **kw)       # a loose bound may pick another (equally valid) arithmetic form
    assert float((a[0] - c[0]).abs().max()) < 2e-6
    with pytest.raises(RuntimeError, match='planes_absmax'):
        gnerf_hip.render_forward(*args, planes_absmax=amax * 0.5, **kw)
    monkeypatch.delenv('GNERF_VERIFY_ABSMAX')
    gnerf_hip.render_forward(*args, planes_absmax=amax * 0.5, **kw)          # unchecked in production: the caller's responsibility


def test_render_interleaved_plane_layout(dev):
    """The renderer reads the producer's channels_last planes in place: planes as [N,H,W,96] (the three planes of an item
    interleaved per texel = channels_last memory of the backbone's [N,96,H,W] output) give bit-identical forward results to the
    [3N,H,W,32] layout on every forward kernel, the same point queries, and the same gradients (laid out like the planes)."""
    import gnerf_hip
    for S, F in ((48, 48), (96, 96), (40, 0), (130, 100), (150, 20)):            # pipe<1>, pipe<2>, coop, pipe<3>, generic
        planes, dec, o, d, nc, nf = _random_scene(17, N=2, res=8, S=S, F=F, hw=(24, 20))
        N = 2
        sep = gnerf_hip.planes_to_nhwc(planes.to(dev))                                           # [6,24,20,32]
        inter = planes.to(dev).reshape(N, 96, 24, 20).permute(0, 2, 3, 1).contiguous()           # [2,24,20,96]
        args = (N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None)
        kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=8)
        a = gnerf_hip.render_forward(sep, *args, **kw)
        b = gnerf_hip.render_forward(inter, *args, **kw)
        for x, y in zip(a, b):
            assert torch.equal(x, y), (S, F)
        gen = torch.Generator().manual_seed(1)
        g = [torch.randn(N, 64, k, generator=gen).to(dev) for k in (32, 1, 1)]
        gp_s, gd_s = gnerf_hip.render_backward(sep, *args, *g, **kw)
        gp_i, gd_i = gnerf_hip.render_backward(inter, *args, *g, **kw)
        assert gp_i.shape == inter.shape
        back = gp_i.reshape(N, 24, 20, 3, 32).permute(0, 3, 1, 2, 4).reshape(6, 24, 20, 32)
        assert float((back - gp_s).abs().max()) <= 1e-5 * float(gp_s.abs().max())
        for x, y in zip(gd_s, gd_i):
            assert _rel(x, y) < 1e-5
    pts = (torch.rand(2, 77, 3, generator=gen) - 0.5).to(dev) * 1.3
    qa, qb = gnerf_hip.query_points(sep, 2, args[1], pts, 1.0), gnerf_hip.query_points(inter, 2, args[1], pts, 1.0)
    assert torch.equal(qa[0], qb[0]) and torch.equal(qa[1], qb[1])
    gs, gc = torch.randn(2, 77, 1, device=dev), torch.randn(2, 77, 32, device=dev)
    ba, bb = gnerf_hip.query_points_backward(sep, 2, args[1], pts, 1.0, gs, gc), gnerf_hip.query_points_backward(inter, 2, args[1], pts, 1.0, gs, gc)
    back = bb[0].reshape(2, 24, 20, 3, 32).permute(0, 3, 1, 2, 4).reshape(6, 24, 20, 32)
    assert float((back - ba[0]).abs().max()) <= 1e-5 * float(ba[0].abs().max())
    with pytest.raises(RuntimeError):
        gnerf_hip.render_forward(inter[:, :, :, :64].contiguous(), *args, **kw)


def test_upsample2x_add_channels_last_producer(dev):
    """The tri-plane producer's last step as one kernel: upsample2d(img, [1,3,3,1]) + y written channels_last, with max |out|;
    against the numpy oracle of upfirdn2d, in both filter orientations, with and without y; its gradient against autograd
    through the composed ops; shapes it does not cover fall back to the composed ops."""
    import gnerf_hip
    from oracle import ops_ref as O
    from torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(0)
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    f_odd = (f + torch.arange(16, device=dev).reshape(4, 4) * 0.01)                  # asymmetric: orientation matters
    for (n, c, h, w), filt, flip, with_y in [((2, 96, 8, 16), f, False, True), ((1, 32, 6, 32), f_odd, False, True),
                                             ((1, 64, 2, 16), f_odd, True, False), ((4, 96, 128, 128), f, False, True)]:
        img = torch.randn(n, c, h, w, generator=gen).to(dev)
        y = torch.randn(n, c, 2 * h, 2 * w, generator=gen).to(dev) if with_y else None
        out, amax = gnerf_hip.upsample2x_add_nhwc(img, y, filt, flip=flip, gain=4.0, with_absmax=True)
        assert out.shape == (n, c, 2 * h, 2 * w) and out.is_contiguous(memory_format=torch.channels_last)
        ref = O.upfirdn2d(img.cpu().numpy().astype(np.float64), filt.cpu().numpy(), up=2, padding=[2, 1, 2, 1], flip_filter=flip, gain=4.0)
        if with_y:
            ref = ref + y.cpu().numpy()
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5)
        assert float(amax) == float(out.abs().max())
    assert gnerf_hip.upsample2x_add_nhwc(torch.randn(1, 3, 8, 16, device=dev), None, f) is None
    # the public entry: gradients of both inputs equal those of the composed ops; uncovered shapes compose
    img = torch.randn(2, 32, 8, 16, generator=gen).to(dev).requires_grad_(True)
    y = torch.randn(2, 32, 16, 32, generator=gen).to(dev).requires_grad_(True)
    w8 = torch.randn(2, 32, 16, 32, generator=gen).to(dev)
    fused = upfirdn2d.upsample2d_add_channels_last(img, y, f)
    assert fused.is_contiguous(memory_format=torch.channels_last)
    gi, gy = torch.autograd.grad((fused * w8).sum(), [img, y])
    plain = upfirdn2d.upsample2d(img, f) + y
    gi2, gy2 = torch.autograd.grad((plain * w8).sum(), [img, y])
    np.testing.assert_allclose(fused.detach().cpu().numpy(), plain.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gi.cpu().numpy(), gi2.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(gy.cpu().numpy(), gy2.cpu().numpy(), rtol=1e-5, atol=1e-5)
    small = upfirdn2d.upsample2d_add_channels_last(torch.randn(1, 3, 8, 16, device=dev), None, f)
    assert small.shape == (1, 3, 16, 32)


def test_importance_renderer_reads_channels_last_planes_in_place(dev, monkeypatch):
    """Drop-in class: planes that arrive as triplane.py:74's view of a channels_last [N,96,H,W] tensor are rendered without the
    NCHW->NHWC layout change (gnerf_hip.planes_to_nhwc is never called), with the image an NCHW copy of the same planes gives;
    gradients come back laid out like the planes."""
    import gnerf_hip
    import gnerf_harness as H
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from training.volumetric_rendering.ray_sampler import RaySampler
    torch.manual_seed(0)
    dec = H.TriPlaneDecoder().to(dev)
    opts = dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1, clamp_mode='softplus',
                disparity_space_sampling=False)
    img = torch.randn(2, 96, 32, 32, device=dev)
    planes_nchw = img.view(2, 3, 32, 32, 32)
    planes_cl = img.contiguous(memory_format=torch.channels_last).view(2, 3, 32, 32, 32)
    assert not planes_cl.is_contiguous()
    c = torch.cat([H.camera_label(H.orbit_pose(i, 120)) for i in (3, 40)]).to(dev)
    o, d = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 16)
    r = ImportanceRenderer()
    with torch.no_grad():
        torch.manual_seed(5)
        want = r(planes_nchw, dec, o, d, opts)
        calls = []
        real = gnerf_hip.planes_to_nhwc
        monkeypatch.setattr(gnerf_hip, 'planes_to_nhwc', lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        torch.manual_seed(5)
        got = r(planes_cl, dec, o, d, opts)
        assert not calls
        for a, b in zip(want, got):
            assert torch.equal(a, b)
        # a ONE-item batch (every gen_videos frame): `view` reports an arbitrary stride for the size-1 dimension, which must not
        # hide the layout (round 3: it did, and each orbit repacked its planes once)
        one_cl = img[:1].contiguous(memory_format=torch.channels_last).view(1, 3, 32, 32, 32)
        assert one_cl.stride(0) != 96 * 32 * 32 or True
        torch.manual_seed(5); want1 = r(planes_nchw[:1], dec, o[:1], d[:1], opts)
        del calls[:]
        torch.manual_seed(5); got1 = r(one_cl, dec, o[:1], d[:1], opts)
        assert not calls and all(torch.equal(a, b) for a, b in zip(want1, got1))
    # training: gradient of the channels_last planes, laid out like them, equals the NCHW route's
    grads = []
    for pl in (planes_nchw, planes_cl):
        del calls[:]
        pl = pl.detach().requires_grad_(True)
        torch.manual_seed(6)
        rgb, depth, _ = r(pl, dec, o, d, opts)
        g, = torch.autograd.grad(rgb.square().sum() + depth.sum(), [pl])
        grads.append(g)
    assert not calls                                        # (the channels_last pass, last in the loop)
    assert grads[1].stride() == planes_cl.stride()
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * float(grads[0].abs().max())


@pytest.mark.production_path
@pytest.mark.parametrize('S', [48, 96, 144])
def test_pipe_kernel_instantiations_agree(dev, monkeypatch, S):
    """render_kernel_pipe<TP, MLP, FULL>: the instantiation with compile-time sample counts (what a 48+48 / 96+96 / 144+144 call
    runs), the general one (GNERF_PIPE_FULL=0) and the debug route give the same bits for every decoder arithmetic -- what the
    stage dumps show is what production computed.  (Round 3: the colour composite, left to the compiler's FMA contraction, was
    1 ulp apart between the two.)"""
    import gnerf_hip
    planes, dec, o, d, nc, nf = _random_scene(3, N=2, res=8, S=S, F=S, hw=(16, 16))
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    args = (nhwc, 2, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev))
    kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=8)
    for mlp in ('auto', 'f16x3', 'f32'):
        monkeypatch.delenv('GNERF_PIPE_FULL', raising=False)
        full = gnerf_hip.render_forward(*args, mlp=mlp, **kw)
        monkeypatch.setenv('GNERF_PIPE_FULL', '0')
        general = gnerf_hip.render_forward(*args, mlp=mlp, **kw)
        debug = gnerf_hip.render_forward(*args, mlp=mlp, debug=True, **kw)[:3]
        for a, b, c in zip(full, general, debug):
            assert torch.equal(a, b) and torch.equal(b, c), mlp


def test_gpu_fallback_to_pytorch_ops_is_visible(dev):
    """The three GPU calls that leave the fused kernel (rays with a gradient, density_noise > 0 under autograd, a decoder that is not the
    OSGDecoder MLP) run the PyTorch-op form -- and say so with a RuntimeWarning, once per reason (What's weak 9)."""
    import warnings
    import gnerf_harness as H
    from training.volumetric_rendering import renderer as RM
    r = RM.ImportanceRenderer()
    dec = H.TriPlaneDecoder().to(dev).requires_grad_(False)
    planes, _, o, d, _, _ = _random_scene(2, N=1, res=4, S=8, F=8, hw=(8, 8))
    planes, o, d = planes.to(dev), o.to(dev), d.to(dev)
    opts = dict(depth_resolution=8, depth_resolution_importance=8, ray_start=2.25, ray_end=3.3, box_warp=1, clamp_mode='softplus', disparity_space_sampling=False)
    RM._warned_fallbacks.clear()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter('always')
        r(planes, dec, o, d, opts)                                                  # the fused kernel: silent
        assert not [w for w in rec if issubclass(w.category, RuntimeWarning)]
        r(planes, dec, o, d, dict(opts, density_noise=0.5))                         # round 6: inside the fused kernel too -- silent
        assert not [w for w in rec if issubclass(w.category, RuntimeWarning)]
        with torch.enable_grad():                                                   # ... unless a gradient has to flow through it (forward-only option)
            pg = planes.clone().requires_grad_(True)
            r(pg, dec, o, d, dict(opts, density_noise=0.5))
            r(pg, dec, o, d, dict(opts, density_noise=0.5))                         # second time: no second warning
        r(planes, dec, o.clone().requires_grad_(True), d, opts)

        class Other(torch.nn.Module):
            def forward(self, feats, dirs):
                x = feats.mean(1)
                return {'rgb': torch.sigmoid(x), 'sigma': x[..., :1]}
        r(planes, Other(), o, d, opts)
    msgs = [str(w.message) for w in rec if issubclass(w.category, RuntimeWarning)]
    assert len(msgs) == 3 and all('PyTorch-op form' in m for m in msgs), msgs
    assert 'density_noise' in msgs[0] and 'rays need a gradient' in msgs[1] and 'OSGDecoder' in msgs[2]


# ---- in-kernel rays and uniform draws (gnerf_render_params ABI 8; SURVEY section 8a row 1, 8d "in-kernel Philox") -------------------

_RAND_CASES = [(0, (7,)), (123, (1000,)), (5, (4, 16384, 48, 1)), (5, (65536, 48)), (2 ** 40 + 17, (3, 333, 12)), (9, (1, 4096, 96, 1)), (11, (524288 + 3,))]


def test_make_rays_and_draws_equal_the_three_launches(dev):
    """gnerf_make_rays_and_draws (round 6: what bench.py's step runs in front of the render kernel): the rays of gnerf_make_rays and the two
    draws of torch.rand -- same values bit for bit, same generator offset afterwards -- from ONE launch; several sizes, a seed with high
    bits, a draw smaller than ATen's grid and one many times larger."""
    import gnerf_hip
    import gnerf_harness as H
    for seed, n, res, S, F in [(0, 4, 128, 48, 48), (2 ** 40 + 17, 2, 16, 12, 12), (7, 1, 8, 8, 0), (3, 3, 64, 96, 96), (11, 1, 5, 5, 3)]:
        c2w = torch.cat([H.orbit_pose(3 + 5 * i, 120) for i in range(n)]).to(dev)
        intr = torch.tensor(H.FFHQ_INTRINSICS, device=dev).reshape(1, 3, 3).repeat(n, 1, 1)
        torch.cuda.manual_seed(seed)
        torch.rand(5, device=dev)                                              # (an offset that is not zero)
        state = torch.cuda.get_rng_state(dev)
        o0, d0 = gnerf_hip.make_rays(c2w, intr, res)
        a0 = torch.rand([n, res * res, S, 1], device=dev)
        b0 = torch.rand(n * res * res, F, device=dev) if F > 0 else None
        after = int(torch.cuda.default_generators[dev.index].get_offset())
        tail0 = torch.rand(3, device=dev)
        torch.cuda.set_rng_state(state, dev)
        o1, d1, a1, b1 = gnerf_hip.make_rays_and_draws(c2w, intr, res, S, F)
        assert int(torch.cuda.default_generators[dev.index].get_offset()) == after
        assert torch.equal(o0, o1) and torch.equal(d0, d1) and torch.equal(a0, a1) and (b0 is None) == (b1 is None) and (b0 is None or torch.equal(b0, b1))
        assert torch.equal(torch.rand(3, device=dev), tail0)                   # the stream of draws goes on as if torch.rand had been called


def test_philox_restatement_matches_torch_rand(dev):
    """oracle/philox_ref.py -- ATen's uniform kernel over Philox4x32-10, restated in numpy -- against the device generator itself:
    every element of two consecutive torch.rand draws, and the generator's offset after each.  This is the pin of the restatement
    (a device generator only exists on the GPU box); the in-kernel draws are then held to both."""
    from oracle import philox_ref as P
    pr = torch.cuda.get_device_properties(dev)
    mp, mt = pr.multi_processor_count, pr.max_threads_per_multi_processor
    gen = torch.cuda.default_generators[dev.index]
    for seed, shape in _RAND_CASES:
        torch.manual_seed(seed)
        o0 = gen.get_offset()
        a = torch.rand(shape, device=dev)
        o1 = gen.get_offset()
        b = torch.rand(shape, device=dev)
        o2 = gen.get_offset()
        n = a.numel()
        wa, p1 = P.torch_rand(n, gen.initial_seed(), o0, mp, mt)
        wb, p2 = P.torch_rand(n, gen.initial_seed(), o1, mp, mt)
        assert (p1, p2) == (o1, o2), (seed, shape, (o0, o1, o2), (p1, p2))
        assert np.array_equal(a.cpu().numpy().reshape(-1), wa) and np.array_equal(b.cpu().numpy().reshape(-1), wb), (seed, shape)


def test_native_torch_rand_matches_torch_rand(dev):
    """gnerf_torch_rand (the render kernels' draw function as a stand-alone kernel) == torch.rand bit for bit at the same generator
    state, and gnerf_torch_rand_plan predicts the offset torch leaves behind."""
    import gnerf_hip
    gen = torch.cuda.default_generators[dev.index]
    for seed, shape in _RAND_CASES:
        n = int(np.prod(shape))
        for skip in (0, 3):                                     # ... also from a generator that has been used before
            torch.manual_seed(seed)
            for _ in range(skip):
                torch.rand(1000, device=dev)
            o0 = gen.get_offset()
            mine = gnerf_hip.torch_rand(n, dev, gen.initial_seed(), o0)
            assert gen.get_offset() == o0                       # the native draw does not touch the generator
            theirs = torch.rand(shape, device=dev)
            assert torch.equal(mine, theirs.reshape(-1)), (seed, shape, skip)
            assert gen.get_offset() == o0 + gnerf_hip.torch_rand_geometry(n, dev)[1]
            assert float(mine.min()) >= 0.0 and float(mine.max()) < 1.0


@pytest.mark.production_path
@pytest.mark.parametrize('S', [48, 96])
def test_render_with_inkernel_rays_and_draws(dev, S):
    """The render kernel making its rays (from the cameras) and its two uniform draws (torch's Philox stream) itself: outputs equal
    the tensor forms' bit for bit -- rays only, draws only, both -- and the generator ends where the two torch.rand calls leave it."""
    import gnerf_hip
    from oracle import render_ref as R
    N, res = 3, 12
    g = torch.Generator().manual_seed(21)
    planes = (torch.randn(N, 3, 32, 24, 20, generator=g) * 1.5).to(dev)
    dec = [t.to(dev) for t in R.fold_decoder(torch.randn(64, 32, generator=g), torch.randn(64, generator=g) * 0.2,
                                             torch.randn(33, 64, generator=g), torch.randn(33, generator=g) * 0.2)]
    c2w = torch.cat([R.lookat_pose(3.14 / 2 + 0.4 * i, 3.14 / 2 - 0.1 * i, 2.7) for i in range(N)]).to(dev)
    intr = torch.tensor([[4.2647, 0.01, 0.5], [0, 4.1, 0.49], [0, 0, 1]]).repeat(N, 1, 1).to(dev)        # skew and unequal focal lengths on purpose
    nhwc = gnerf_hip.planes_to_nhwc(planes)
    kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)
    gen = torch.cuda.default_generators[dev.index]
    assert gnerf_hip.render_generated_supported(S, S)
    o, d = gnerf_hip.make_rays(c2w, intr, res)
    for mlp in ('auto', 'f32'):
        torch.manual_seed(77)
        torch.rand(5, device=dev)                                   # a generator that is not at offset 0
        nc = torch.rand([N, res * res, S, 1], device=dev)
        nf = torch.rand(N * res * res, S, device=dev)
        end = gen.get_offset()
        want = gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, mlp=mlp, **kw)
        # rays in the kernel, draws as tensors
        got = gnerf_hip.render_forward(nhwc, N, dec, None, None, nc, nf, mlp=mlp, cameras=(c2w, intr, res), **kw)
        for a, b in zip(got, want):
            assert torch.equal(a, b), ('rays', mlp)
        # draws in the kernel (generator put back where the tensor draws started), rays as tensors -- then both
        for cams in (None, (c2w, intr, res)):
            torch.manual_seed(77)
            torch.rand(5, device=dev)
            plan = gnerf_hip.torch_philox_plan(dev, N, res * res, S, S)
            assert gen.get_offset() == end == plan.end_offset
            got = gnerf_hip.render_forward(nhwc, N, dec, None if cams else o, None if cams else d, None, None, mlp=mlp, cameras=cams, rng=plan, **kw)
            for a, b in zip(got, want):
                assert torch.equal(a, b), ('draws', cams is not None, mlp)
    # per-item draws: what N calls of one item each would draw (the batched-views form of an orbit), one shared set of planes
    one = gnerf_hip.planes_to_nhwc(planes[:1])
    torch.manual_seed(78)
    draws = [(torch.rand([1, res * res, S, 1], device=dev), torch.rand(res * res, S, device=dev)) for _ in range(N)]
    end = gen.get_offset()
    want = gnerf_hip.render_forward(one, N, dec, o, d, torch.cat([c for c, _ in draws]), torch.cat([f for _, f in draws]),
                                    planes_shared=True, depth_clamp_per_item=True, **kw)
    torch.manual_seed(78)
    plan = gnerf_hip.torch_philox_plan(dev, N, res * res, S, S, per_item=True)
    assert gen.get_offset() == end
    got = gnerf_hip.render_forward(one, N, dec, None, None, None, None, planes_shared=True, depth_clamp_per_item=True, cameras=(c2w, intr, res), rng=plan, **kw)
    for a, b in zip(got, want):
        assert torch.equal(a, b), 'per-item draws'
    # shapes the generating instantiations are not built for are refused (the caller passes tensors), never rendered differently
    with pytest.raises(RuntimeError):
        gnerf_hip.render_forward(nhwc, N, dec, None, None, None, None, cameras=(c2w, intr, res), rng=gnerf_hip.torch_philox_plan(dev, N, res * res, 40, 40, advance=False),
                                 **dict(kw, depth_resolution=40, depth_resolution_importance=40))


@pytest.mark.production_path
def test_views_of_one_item_equal_separate_calls(dev):
    """Frame batching (an orbit's frames are N cameras on ONE latent's planes): one launch over N views of one set of planes --
    gnerf_render_params.planes_shared + depth_clamp_per_item -- gives every view bit-identically what a launch of its own
    gives, including the final depth clamp (ray_marcher.py:49-50 takes the range over the whole call, i.e. over ONE frame in
    gen_videos.py), through both bindings, both plane layouts, the kernels for full-size and odd sample counts, and the
    renderer class (which also makes the uniform draws in the order N calls would)."""
    import gnerf_hip
    import gnerf_harness as H
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from training.volumetric_rendering.ray_sampler import RaySampler
    torch.manual_seed(0)
    dec = H.TriPlaneDecoder().to(dev)
    r = ImportanceRenderer()
    fcs = (dec.net[0], dec.net[2])
    N, res = 5, 20                                                       # 400 rays per view: workgroups straddle views
    c = torch.cat([H.camera_label(H.orbit_pose(i, 120)) for i in (0, 17, 33, 61, 95)]).to(dev)
    o, d = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), res)
    img = torch.randn(1, 96, 32, 32, device=dev) * 0.5
    with torch.no_grad():
        w = r._decoder_cache(fcs)
        for S, F in ((48, 48), (96, 96), (13, 7), (24, 0)):
            for planes in (gnerf_hip.planes_to_nhwc(img.view(1, 3, 32, 32, 32)), img.permute(0, 2, 3, 1).contiguous()):
                nc, nf = torch.rand(N * res * res, S, device=dev), (torch.rand(N * res * res, F, device=dev) if F else None)
                kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)
                for binding in ('ext', 'ctypes'):
                    old = gnerf_hip._ext
                    try:
                        if binding == 'ctypes':
                            gnerf_hip._ext = False
                        got = gnerf_hip.render_forward(planes, N, w, o, d, nc, nf, planes_shared=True, depth_clamp_per_item=True, **kw)
                        # the same launch with the planes copied N times and a per-item clamp: the shared read changes nothing
                        rep = gnerf_hip.render_forward(planes.repeat(N, 1, 1, 1), N, w, o, d, nc, nf, depth_clamp_per_item=True, **kw)
                        whole = gnerf_hip.render_forward(planes.repeat(N, 1, 1, 1), N, w, o, d, nc, nf, **kw)
                        for i in range(N):
                            sl = slice(i * res * res, (i + 1) * res * res)
                            one = gnerf_hip.render_forward(planes, 1, w, o[i:i + 1], d[i:i + 1], nc[sl], None if nf is None else nf[sl], **kw)
                            for a, b, e in zip(got, rep, one):
                                assert torch.equal(a[i:i + 1], e) and torch.equal(b[i:i + 1], e), (S, F, binding, i)
                        # ... and the call-wide clamp is a different thing (or the test above shows nothing): colours equal, depths
                        # clamped to the union of the ranges
                        assert torch.equal(whole[0], got[0]) and torch.equal(whole[2], got[2])
                        assert float(whole[1].min()) <= float(got[1].min()) and float(whole[1].max()) >= float(got[1].max())
                    finally:
                        gnerf_hip._ext = old
        # the workspace is back to idle (all zero): the next call-wide clamp starts from nothing
        torch.cuda.synchronize()
        for ws in gnerf_hip._workspaces.values():
            words = ws.view(torch.int32)
            assert int(words[:4].abs().sum()) == 0 and int(words[16:].abs().sum()) == 0
        # the renderer class: planes of one item, rays of N
        opts = dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1, clamp_mode='softplus',
                    disparity_space_sampling=False)
        planes5 = img.view(1, 3, 32, 32, 32)
        torch.manual_seed(9)
        got = r(planes5, dec, o, d, opts)
        torch.manual_seed(9)
        for i in range(N):
            one = r(planes5, dec, o[i:i + 1], d[i:i + 1], opts)
            for a, e in zip(got, one):
                assert torch.equal(a[i:i + 1], e)
        # 'auto' ray limits (renderer.py:91-96) are per call as well
        opts_auto = dict(opts, ray_start='auto', ray_end='auto', box_warp=2.0)
        torch.manual_seed(10)
        got = r(planes5, dec, o, d, opts_auto)
        torch.manual_seed(10)
        for i in range(N):
            one = r(planes5, dec, o[i:i + 1], d[i:i + 1], opts_auto)
            for a, e in zip(got, one):
                assert torch.equal(a[i:i + 1], e)
    # the several-views LAUNCH is forward-only; with a graph the class makes one differentiable call per view: same values as the
    # views launch on the same draws, and a plane gradient that is the sum over the views
    pg = planes5.clone().requires_grad_(True)
    torch.manual_seed(9)
    with torch.enable_grad():
        outs = r(pg, dec, o, d, opts)
        (gp,) = torch.autograd.grad(outs[0].sum() + outs[1].sum(), pg)
    torch.manual_seed(9)
    want = r(planes5, dec, o, d, opts)
    for a, e in zip(outs, want):
        assert torch.equal(a.detach(), e)
    acc = torch.zeros_like(gp)
    torch.manual_seed(9)
    for i in range(N):
        pi = planes5.clone().requires_grad_(True)
        with torch.enable_grad():
            one = r(pi, dec, o[i:i + 1], d[i:i + 1], opts)
            acc += torch.autograd.grad(one[0].sum() + one[1].sum(), pi)[0]
    assert float((gp - acc).abs().max()) <= 1e-4 * float(acc.abs().max())
    nhwc = gnerf_hip.planes_to_nhwc(planes5)
    p, keep, m = gnerf_hip._render_params(nhwc, N, w, o, d, torch.rand(N * res * res, 48, device=dev), torch.rand(N * res * res, 48, device=dev),
                                          48, 48, 2.25, 3.3, 1.0, False, False, res, 'render_backward', planes_shared=True)
    g = gnerf_hip.RenderGrads()
    assert gnerf_hip.load().gnerf_render_backward(ctypes.byref(p), ctypes.byref(g), None) == gnerf_hip.E_UNSUPPORTED


def test_orbit_frames_per_call(dev):
    """gen_videos orbit with k cameras per synthesis call: the renderer's share of every frame (the depth image, triplane.py:78)
    is bit-identical to the frame-at-a-time orbit when both draw the same uniforms; the images that went through the
    superresolution's fp16 convolutions agree to their noise (MIOpen picks its algorithm per batch size)."""
    import gnerf_harness as H
    import gen_videos_mi355x as GV
    G = GV.build_random_generator(3, dev)
    z = torch.randn(1, G.z_dim, device=dev)
    outs = {}
    for k in (1, 4):
        torch.manual_seed(21)
        frames, raws, (lo, hi) = GV.render_orbit(G, z, 10, 32, dev, double_depth=False, frames_per_call=k)
        assert frames.shape == (10, 512, 512, 3) and raws.shape == (10, 32, 32, 3) and (lo, hi) == (0, 10)
        outs[k] = (frames, raws)
    for which in (0, 1):
        diff = (outs[1][which].int() - outs[4][which].int()).abs()
        assert int(diff.max()) <= 16 and float(diff.float().mean()) < 0.5, (which, int(diff.max()), float(diff.float().mean()))
    with torch.no_grad():
        ws = GV.orbit_latents(G, z, dev)
        cams = torch.cat([H.camera_label(H.orbit_pose(i, 10, G.rendering_kwargs['avg_camera_radius'])) for i in range(4)]).to(dev)
        G.synthesis(ws=ws, c=cams[:1], noise_mode='const', neural_rendering_resolution=32, cache_backbone=True)
        torch.manual_seed(22)
        together = G.synthesis(ws=ws, c=cams, noise_mode='const', neural_rendering_resolution=32, use_cached_backbone=True)
        assert together['image'].shape == (4, 3, 512, 512)
        torch.manual_seed(22)
        for i in range(4):
            alone = G.synthesis(ws=ws, c=cams[i:i + 1], noise_mode='const', neural_rendering_resolution=32, use_cached_backbone=True)
            assert torch.equal(alone['image_depth'], together['image_depth'][i:i + 1])
    # a graph program of 4 views, 10 frames (the last block is short)
    torch.manual_seed(21)
    fg, rg, _ = GV.render_orbit(G, z, 10, 32, dev, double_depth=False, frames_per_call=4, use_graph=True)
    assert fg.shape == (10, 512, 512, 3)
    assert float((fg.int() - outs[1][0].int()).abs().float().mean()) < 1.0         # other draws: same picture up to sampling noise
    with pytest.raises(ValueError):
        GV.render_orbit(G, torch.randn(2, G.z_dim, device=dev), 4, 32, dev, double_depth=False, frames_per_call=2)


def test_one_latent_folds_input_scaling_into_packed_weights(dev, monkeypatch):
    """An orbit renders k views of ONE latent: the layers whose input scaling no earlier epilogue carries (the first layer of a superresolution
    block) then fold it into their packed weights, per latent (gnerf_generator._packed_modulated_weight: the order the reference's fused
    inference form takes, networks_stylegan2.py:66-75), instead of a scaling pass over the activations.  The route is taken (fewer
    gnerf_scale_channels launches, the packed weights cached: a second call packs nothing), GNERF_LATENT_WEIGHTS=0 gives the same image to the
    fp16 layers' rounding, and the renderer's share of the frame does not change at all."""
    import gnerf_generator as GG
    import gnerf_harness as H
    import gen_videos_mi355x as GV
    import gnerf_hip
    G = GV.build_random_generator(3, dev)
    z = torch.randn(1, G.z_dim, device=dev)
    calls = {'scale': 0, 'pack': 0}
    real_scale, real_pack = gnerf_hip.scale_channels, GG._packed_modulated_weight
    monkeypatch.setattr(gnerf_hip, 'scale_channels', lambda *a, **k: (calls.__setitem__('scale', calls['scale'] + 1), real_scale(*a, **k))[1])
    monkeypatch.setattr(GG, '_packed_modulated_weight', lambda *a, **k: (calls.__setitem__('pack', calls['pack'] + 1), real_pack(*a, **k))[1])
    with torch.no_grad():
        ws = GV.orbit_latents(G, z, dev)
        cams = torch.cat([H.camera_label(H.orbit_pose(i, 10, G.rendering_kwargs['avg_camera_radius'])) for i in range(4)]).to(dev)
        G.synthesis(ws=ws, c=cams[:1], noise_mode='const', neural_rendering_resolution=32, cache_backbone=True)
        out = {}
        for on in (True, False, True):
            monkeypatch.setattr(GG, '_LATENT_WEIGHTS', on)
            calls['scale'] = calls['pack'] = 0
            torch.manual_seed(3)
            o = G.synthesis(ws=ws, c=cams, noise_mode='const', neural_rendering_resolution=32, use_cached_backbone=True)
            out.setdefault(on, (o, dict(calls)))
            last = dict(calls)
    (img_on, c_on), (img_off, c_off) = out[True], out[False]
    assert c_off['pack'] == 0 and c_on['scale'] < c_off['scale'], (c_on, c_off)
    assert last['pack'] == 0                                             # the third call found every packed weight in the per-latent cache
    assert torch.equal(img_on['image_depth'], img_off['image_depth']) and torch.equal(img_on['image_raw'], img_off['image_raw'])
    d = (img_on['image'].float() - img_off['image'].float()).abs()
    scale = float(img_off['image'].float().abs().max())
    assert float(d.max()) < 0.03 * max(scale, 1.0) and float(d.mean()) < 2e-3 * max(scale, 1.0), (float(d.max()), float(d.mean()), scale)


def test_to_uint8_nhwc_matches_the_pytorch_ops(dev):
    """gnerf_to_uint8_nhwc == (img * 127.5 + 128).clamp(0, 255).to(uint8).permute(0, 2, 3, 1) (gen_videos.py:173), byte for byte: random
    images, values on and around every half-integer boundary of the product, +-inf, out-of-range values; NaN -> 0."""
    import gnerf_hip
    import gnerf_harness as H
    gen = torch.Generator().manual_seed(11)
    for shape in ((1, 3, 512, 512), (4, 3, 64, 64), (2, 1, 5, 7), (1, 32, 9, 3)):
        img = (torch.randn(*shape, generator=gen) * 0.7).to(dev)
        want = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        got = gnerf_hip.to_uint8_nhwc(img)
        assert got.dtype == torch.uint8 and got.shape == want.shape and torch.equal(got, want)
        assert torch.equal(H.to_uint8(img), want)
    edges = torch.cat([(torch.arange(-2, 258, dtype=torch.float64) - 128) / 127.5 + d for d in (-1e-6, 0.0, 1e-6)]).float()
    edges = torch.cat([edges, torch.tensor([float('inf'), -float('inf'), 1e30, -1e30, 3.0, -3.0])])
    img = edges.reshape(1, 1, 1, -1).to(dev)
    want = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    assert torch.equal(gnerf_hip.to_uint8_nhwc(img), want)
    nan = torch.full((1, 3, 2, 2), float('nan'), device=dev)
    assert int(gnerf_hip.to_uint8_nhwc(nan).max()) == 0
    with pytest.raises(RuntimeError):
        gnerf_hip.to_uint8_nhwc(img.half())


def test_query_points_vs_oracle(dev):
    import gnerf_hip
    from oracle import render_ref as R
    planes, dec, *_ = _random_scene(5, N=2, res=4, S=4, F=0, hw=(12, 10))
    gen = torch.Generator().manual_seed(2)
    pts = (torch.rand(2, 77, 3, generator=gen) - 0.5) * 1.3
    ref_sigma, ref_rgb = R.query_points(planes, dec, pts, 1.0)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    sigma, rgb = gnerf_hip.query_points(nhwc, 2, [t.to(dev) for t in dec], pts.to(dev), 1.0)
    np.testing.assert_allclose(sigma.cpu().numpy(), ref_sigma.numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(rgb.cpu().numpy(), ref_rgb.numpy(), rtol=0, atol=2e-5)


def test_importance_renderer_dropin_on_gpu(dev, golden):
    """The drop-in class on GPU tensors: takes the fused path, consumes the torch generator exactly like the
    PyTorch-op path (same two draws), and matches it."""
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from test_host_cpu import Decoder, options_of
    g = golden('render_s48.npz')
    ren = ImportanceRenderer().to(dev)
    dec = Decoder(g).to(dev)
    planes, o, d = _t(g['planes'], dev), _t(g['ray_origins'], dev), _t(g['ray_dirs'], dev)
    opts = options_of(g)
    with torch.no_grad():
        torch.manual_seed(123)
        hip = ren(planes, dec, o, d, opts)
        state_after_hip = torch.cuda.get_rng_state(dev)
        torch.manual_seed(123)
        ref = ren._forward_torch(planes, dec, o, d, opts)
        state_after_ref = torch.cuda.get_rng_state(dev)
    assert torch.equal(state_after_hip, state_after_ref)
    assert float(((hip[0] - ref[0]) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(hip[1].cpu().numpy(), ref[1].cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(hip[2].cpu().numpy(), ref[2].cpu().numpy(), atol=2e-4)
    assert '_gnerf_planes_cache' in ren.__dict__          # proves the fused path ran
    # run_model (TriPlaneGenerator.sample / sample_mixed entry point)
    pts = (torch.rand(1, 50, 3, device=dev) - 0.5)
    with torch.no_grad():
        out = ren.run_model(planes, dec, pts, torch.zeros_like(pts), opts)
        from training.volumetric_rendering.renderer import sample_from_planes
        ref_out = dec(sample_from_planes(ren.plane_axes, planes, pts, padding_mode='zeros', box_warp=opts['box_warp']), None)
    np.testing.assert_allclose(out['sigma'].cpu().numpy(), ref_out['sigma'].cpu().numpy(), rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(out['rgb'].cpu().numpy(), ref_out['rgb'].cpu().numpy(), atol=2e-5)


# ---------------------------------------------------------------------------- renderer gradient


def _oracle_grads(planes, dec, o, d, nc, nf, opts, g_rgb, g_depth, g_wsum, dtype=torch.float64):
    """Autograd through the CPU oracle (float64 by default): the reference's backward is autograd through the same ops."""
    from oracle import render_ref as R
    pl = planes.to(dtype).requires_grad_(True)
    dc = [t.to(dtype).requires_grad_(True) for t in dec]
    rgb, depth, w = R.render(pl, dc, o.to(dtype), d.to(dtype), opts, nc.to(dtype), nf.to(dtype))
    loss = (rgb * g_rgb.to(dtype)).sum() + (depth * g_depth.to(dtype)).sum() + (w * g_wsum.to(dtype)).sum()
    loss.backward()
    return pl.grad, [t.grad for t in dc]


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _rel_l2(a, b):
    """|a - b|_2 / |b|_2: unlike the max-entry ratio this sees errors spread over the many small entries of a gradient."""
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _entrywise_ok(a, b, rtol=5e-3, floor=2e-4):
    """Every entry within rtol of its own magnitude, with an absolute floor of `floor` x the largest entry (atomics in
    arbitrary order and fp32 arithmetic against a float64 reference)."""
    a, b = a.double(), b.double()
    return bool(((a - b).abs() <= rtol * b.abs() + floor * b.abs().max()).all())


@pytest.mark.parametrize('cfg', [
    dict(N=2, res=8, S=48, F=48, hw=(32, 32)),              # the training sample counts
    dict(N=1, res=5, S=17, F=30, hw=(9, 11)),               # ragged: partial tiles, odd ray count (idle waves in the last workgroup)
    dict(N=1, res=4, S=12, F=0, hw=(8, 8)),                 # no importance pass (round 6: on the pipelined path like the others)
    dict(N=2, res=8, S=48, F=0, hw=(16, 16)),               # ... at the training sample count, whole ray tiles
    dict(N=1, res=5, S=100, F=0, hw=(8, 12)),               # ... seven coarse tiles (three per shader wave), ragged ray tiles
    dict(N=1, res=4, S=96, F=96, hw=(16, 16), white_back=True),
    dict(N=3, res=4, S=4, F=5, hw=(4, 4)),
    dict(N=3, res=5, S=20, F=24, hw=(12, 10)),              # ragged ITEMS: 25 rays each, no image tiling -- round 6: items padded to whole ray tiles on the staged route
    dict(N=1, res=3, S=130, F=100, hw=(8, 8)),              # beyond 96+96, ragged tiles, 9 rays: partial workgroup
    dict(N=1, res=2, S=256, F=256, hw=(8, 8)),              # GNERF_MAX_SAMPLES on both passes (159 KB of LDS per workgroup)
])
def test_render_backward_vs_oracle(dev, cfg, monkeypatch):
    """gnerf_render_backward against autograd through the fp64 oracle: plane, weight and bias gradients of a random
    linear functional of (rgb, depth, weight sum).  Tolerance: 2e-3 of each gradient's largest entry (fp32 kernel,
    hardware exp/log, atomics in arbitrary order, vs a float64 reference)."""
    import gnerf_hip
    cfg = dict(cfg)
    white_back = cfg.pop('white_back', False)
    planes, dec, o, d, nc, nf = _random_scene(11, **cfg)
    S, F, N = cfg['S'], cfg['F'], cfg['N']
    M = cfg['res'] ** 2
    gen = torch.Generator().manual_seed(5)
    g_rgb = torch.randn(N, M, 32, generator=gen)
    g_depth = torch.randn(N, M, 1, generator=gen)
    g_wsum = torch.randn(N, M, 1, generator=gen)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus',
                white_back=white_back)
    ref_planes, ref_dec = _oracle_grads(planes, dec, o, d, nc, nf, opts, g_rgb, g_depth, g_wsum)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    # staged, pipelined first pass + sample-tile kernel (the default where the pipelined kernels cover the sample counts) / staged,
    # one wave per ray (GNERF_BWD_KERNEL=wave; what every other shape runs) / single pass, one atomic per tap and channel
    # (GNERF_BWD_MLP=f32: the pipelined path with exact-fp32 products in BOTH kernels -- each otherwise picks f16 hi/lo or fp32 on the
    # device like the forward)
    # (round 5: the staged form's second pass bins the rows by plane tile and sums each tile in LDS -- no float atomics;
    # GNERF_BWD_SCATTER=sorted keeps round 2's per-ray-tile sort with one atomic per texel and chunk)
    for staged, kernel in ((True, None), (True, 'f32'), (True, 'wave'), (True, 'sorted'), (False, None)):
        monkeypatch.delenv('GNERF_BWD_KERNEL', raising=False)
        monkeypatch.delenv('GNERF_BWD_MLP', raising=False)
        monkeypatch.delenv('GNERF_BWD_SCATTER', raising=False)
        if kernel == 'wave':
            monkeypatch.setenv('GNERF_BWD_KERNEL', kernel)
        elif kernel == 'f32':
            monkeypatch.setenv('GNERF_BWD_MLP', 'f32')
        elif kernel == 'sorted':
            monkeypatch.setenv('GNERF_BWD_SCATTER', 'sorted')
        gp, gdec = gnerf_hip.render_backward(nhwc, N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None,
                                             g_rgb.to(dev), g_depth.to(dev), g_wsum.to(dev),
                                             depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                                             white_back=white_back, image_width=cfg['res'], staged_scatter=staged)
        gp_nchw = gp.reshape(N, 3, *cfg['hw'], 32).permute(0, 1, 4, 2, 3).cpu()
        assert _rel(gp_nchw, ref_planes) < 2e-3, (staged, kernel, _rel(gp_nchw, ref_planes))
        assert _rel_l2(gp_nchw, ref_planes) < 1e-3, (staged, kernel, _rel_l2(gp_nchw, ref_planes))
        assert _entrywise_ok(gp_nchw, ref_planes), (staged, kernel)
        for name, a, b in zip(['w1', 'b1', 'w2', 'b2'], gdec, ref_dec):
            assert _rel(a.cpu(), b) < 2e-3, (name, kernel, _rel(a.cpu(), b))
            assert _rel_l2(a.cpu(), b) < 1e-3, (name, kernel, _rel_l2(a.cpu(), b))
    monkeypatch.delenv('GNERF_BWD_KERNEL', raising=False)
    monkeypatch.delenv('GNERF_BWD_MLP', raising=False)
    monkeypatch.delenv('GNERF_BWD_SCATTER', raising=False)
    # each input gradient alone (NULL pointers for the others) and planes-only / decoder-only requests
    gp2, none_dec = gnerf_hip.render_backward(nhwc, N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None,
                                              g_rgb.to(dev), None, None, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3,
                                              box_warp=1.0, white_back=white_back, image_width=0, need_decoder=False)
    assert none_dec is None
    ref_planes2, _ = _oracle_grads(planes, dec, o, d, nc, nf, opts, g_rgb, torch.zeros_like(g_depth), torch.zeros_like(g_wsum))
    assert _rel(gp2.reshape(N, 3, *cfg['hw'], 32).permute(0, 1, 4, 2, 3).cpu(), ref_planes2) < 2e-3


def test_render_backward_ragged_items_take_the_staged_route(dev, monkeypatch):
    """Several items whose ray count is no multiple of 16 and no image tiling: until round 6 such a call left the staged route for the
    one-wave-per-ray kernel with float atomics (a 16-ray tile would straddle items).  The ray sequence now pads every item to whole tiles:
    GNERF_BWD_SCATTER=staged -- which REFUSES a call the staged route cannot take -- succeeds, the result agrees with the fp64 oracle, and it is
    bit-identical from run to run (the binned scatter's property)."""
    import gnerf_hip
    N, res, S, F, hw = 4, 7, 48, 48, (24, 20)
    planes, dec, o, d, nc, nf = _random_scene(23, N=N, res=res, S=S, F=F, hw=hw)
    M = res * res
    gen = torch.Generator().manual_seed(8)
    g_rgb, g_depth, g_wsum = torch.randn(N, M, 32, generator=gen), torch.randn(N, M, 1, generator=gen), torch.randn(N, M, 1, generator=gen)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus', white_back=False)
    ref_planes, ref_dec = _oracle_grads(planes, dec, o, d, nc, nf, opts, g_rgb, g_depth, g_wsum)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    monkeypatch.setenv('GNERF_BWD_SCATTER', 'staged')
    runs = []
    for _ in range(2):
        gp, gdec = gnerf_hip.render_backward(nhwc, N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev), g_rgb.to(dev), g_depth.to(dev), g_wsum.to(dev),
                                             depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=0, staged_scatter=True)
        runs.append(gp)
    assert torch.equal(runs[0], runs[1])
    gp_nchw = runs[0].reshape(N, 3, *hw, 32).permute(0, 1, 4, 2, 3).cpu()
    assert _rel(gp_nchw, ref_planes) < 2e-3 and _rel_l2(gp_nchw, ref_planes) < 1e-3 and _entrywise_ok(gp_nchw, ref_planes)
    for name, a, b in zip(['w1', 'b1', 'w2', 'b2'], gdec, ref_dec):
        assert _rel(a.cpu(), b) < 2e-3 and _rel_l2(a.cpu(), b) < 1e-3, name


def test_render_backward_f16_tile_kernel_agrees_with_fp32_on_every_run(dev, monkeypatch):
    """The backward tile kernel's f16 hi/lo form with two waves per SIMD, forced on (GNERF_BWD_MLP_K2=f16x3), against the all-fp32 pair
    on 36 different incoming gradients spanning twelve orders of magnitude: every plane gradient within 1e-5 of the fp32 one's largest
    entry, every run.  (The form lost 16 dO entries of a handful of sample tiles per launch -- ~1e-3 here, on most launches -- until the
    build started rewriting the packed-fp32 instruction that reads 0.0 next to v_mfma_f32_16x16x32_f16: tests/test_isa_cpu.py.)"""
    import gnerf_hip
    import gnerf_harness as H
    torch.manual_seed(0)
    N, res, S = 2, 32, 48
    M = res * res
    planes = torch.randn(N, 3, 32, 64, 64, device=dev)
    dec = [torch.randn(64, 32, device=dev) * 0.18, torch.randn(64, device=dev) * 0.1, torch.randn(33, 64, device=dev) * 0.12, torch.randn(33, device=dev) * 0.1]
    c2w = torch.cat([H.lookat_pose(3.14 / 2 + 0.3 * i, 3.14 / 2 - 0.05, 2.7) for i in range(N)]).to(dev)
    intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1).to(dev)
    o, d = gnerf_hip.make_rays(c2w, intr, res)
    nc, nf = torch.rand(N * M, S, device=dev), torch.rand(N * M, S, device=dev)
    nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
    kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res, planes_absmax=amax)
    g_rgb, g_depth, g_w = torch.randn(N, M, 32, device=dev), torch.randn(N, M, 1, device=dev), torch.zeros(N, M, 1, device=dev)

    def run(k1, k2, *g):
        monkeypatch.setenv('GNERF_BWD_MLP_K1', k1)
        monkeypatch.setenv('GNERF_BWD_MLP_K2', k2)
        return gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, *g, **kw)
    worst = 0.0
    for i in range(36):
        sc = [1.0, 1e-6, 3.0, 1e6][i % 4]
        g = (g_rgb * sc, torch.zeros_like(g_depth) if i % 2 else g_depth * sc, g_w)
        ref_p, ref_d = run('f32', 'f32', *g)
        out_p, out_d = run('f32', 'f16x3', *g)
        e = float((out_p - ref_p).abs().max() / ref_p.abs().max())
        worst = max(worst, e)
        assert e < 1e-5, (i, sc, e)
        for a, b in zip(out_d, ref_d):
            assert _rel(a, b) < 1e-4, (i, sc, _rel(a, b))
    monkeypatch.delenv('GNERF_BWD_MLP_K1', raising=False)
    monkeypatch.delenv('GNERF_BWD_MLP_K2', raising=False)
    # the default picks the f16 form here (features, weights and activations are inside f16's range): same bound against fp32
    out_p, _ = gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, **kw)
    ref_p, _ = run('f32', 'f32', g_rgb, g_depth, g_w)
    assert float((out_p - ref_p).abs().max() / ref_p.abs().max()) < 1e-5


@pytest.mark.production_path
@pytest.mark.parametrize('layout', ['planes', 'interleaved'])
def test_plane_gradients_are_bit_reproducible(dev, monkeypatch, layout):
    """The binned plane-gradient scatter (csrc/scatter_binned.inl) sums every plane tile in 64-bit fixed point in LDS: integer addition
    is associative, so the plane gradient is the same BITS on every run -- whatever order the records were binned and added in -- which
    neither the float-atomic forms nor the reference's grid_sampler_2d_backward give.  Three runs with identical bits, in both plane
    layouts; against the float-atomic form (GNERF_BWD_SCATTER=sorted) within that form's own run-to-run spread; samples outside the
    planes (zero-padding taps) and a plane size that is no multiple of the 16-texel tile are part of the scene; the call ADDS to what the
    gradient buffer holds (the contract of the atomic forms)."""
    import gnerf_hip
    import gnerf_harness as H
    torch.manual_seed(4)
    N, res, S = 2, 32, 48
    M = res * res
    hh, ww = 72, 56                                                    # 4.5 x 3.5 tiles: partial tiles on both axes
    planes = torch.randn(N, 3, 32, hh, ww, device=dev)
    dec = [torch.randn(64, 32, device=dev) * 0.18, torch.randn(64, device=dev) * 0.1, torch.randn(33, 64, device=dev) * 0.12, torch.randn(33, device=dev) * 0.1]
    c2w = torch.cat([H.lookat_pose(3.14 / 2 + 0.4 * i, 3.14 / 2 - 0.1, 2.7) for i in range(N)]).to(dev)
    intr = torch.tensor([[2.2, 0, 0.5], [0, 2.2, 0.5], [0, 0, 1]]).repeat(N, 1, 1).to(dev)       # a wide field of view: many samples leave the box
    o, d = gnerf_hip.make_rays(c2w, intr, res)
    nc, nf = torch.rand(N * M, S, device=dev), torch.rand(N * M, S, device=dev)
    if layout == 'planes':
        nhwc = gnerf_hip.planes_to_nhwc(planes)
        to_nchw = lambda g: g.reshape(N, 3, hh, ww, 32).permute(0, 1, 4, 2, 3)
    else:
        nhwc = planes.reshape(N, 96, hh, ww).permute(0, 2, 3, 1).contiguous()       # [N,H,W,96]: the backbone's channels_last output
        to_nchw = lambda g: g.reshape(N, hh, ww, 3, 32).permute(0, 3, 4, 1, 2)
    kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)
    g_rgb, g_depth, g_w = torch.randn(N, M, 32, device=dev), torch.randn(N, M, 1, device=dev), torch.randn(N, M, 1, device=dev)
    monkeypatch.delenv('GNERF_BWD_SCATTER', raising=False)
    runs = [gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, need_decoder=False, **kw)[0] for _ in range(3)]
    assert float(runs[0].abs().max()) > 0 and torch.isfinite(runs[0]).all()
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    monkeypatch.setenv('GNERF_BWD_SCATTER', 'sorted')
    atom = [gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, need_decoder=False, **kw)[0] for _ in range(2)]
    monkeypatch.delenv('GNERF_BWD_SCATTER', raising=False)
    top = float(atom[0].abs().max())
    assert float((runs[0] - atom[0]).abs().max()) <= 2e-6 * top, float((runs[0] - atom[0]).abs().max()) / top
    assert _rel_l2(to_nchw(runs[0]), to_nchw(atom[0])) < 1e-6
    # the two layouts hold the same numbers (checked against the other layout's result through the float-atomic form above; here:
    # every texel that receives nothing is exactly zero in both forms)
    assert torch.equal(runs[0] == 0, atom[0] == 0) or float(((runs[0] == 0) != (atom[0] == 0)).float().mean()) < 1e-4


def _oracle_free_planes_grad(results, g_rgb, ren, dec, g, o, d, opts, dev):
    """planes gradient of sum(rgb * g_rgb) through the PyTorch-op path, same seed."""
    planes = _t(g['planes'], dev).requires_grad_(True)
    torch.manual_seed(321)
    out = ren._forward_torch(planes, dec, o, d, opts)
    (out[0] * g_rgb).sum().backward()
    return planes.grad


@pytest.mark.parametrize('case', ['render_s48.npz', 'render_misc.npz'])
def test_importance_renderer_training_step_on_gpu(dev, golden, case):
    """Drop-in class with gradients enabled: the fused forward + backward kernels give the same outputs and the same
    plane / decoder-parameter gradients as autograd through the PyTorch-op path (the reference's behaviour), for the same seed."""
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from test_host_cpu import Decoder, options_of
    g = golden(case)
    ren = ImportanceRenderer().to(dev)
    dec = Decoder(g).to(dev).requires_grad_(True)
    o, d = _t(g['ray_origins'], dev), _t(g['ray_dirs'], dev)
    opts = options_of(g)
    gen = torch.Generator().manual_seed(3)
    N, M = o.shape[:2]
    g_rgb, g_depth, g_w = [torch.randn(N, M, c, generator=gen).to(dev) for c in (32, 1, 1)]
    results = []
    for fused in (True, False):
        planes = _t(g['planes'], dev).requires_grad_(True)
        dec.zero_grad(set_to_none=True)
        torch.manual_seed(321)
        out = ren(planes, dec, o, d, opts) if fused else ren._forward_torch(planes, dec, o, d, opts)
        assert (out[0].grad_fn is not None)
        if fused:
            assert type(out[0].grad_fn).__name__.startswith('_FusedRender')         # proves the fused op is in the graph
        ((out[0] * g_rgb).sum() + (out[1] * g_depth).sum() + (out[2] * g_w).sum()).backward()
        results.append((out, planes.grad.clone(), [p.grad.clone() for p in dec.parameters()]))
    (out_f, gp_f, gd_f), (out_r, gp_r, gd_r) = results
    assert float(((out_f[0] - out_r[0]).detach() ** 2).mean()) < 1e-8
    assert _rel(gp_f, gp_r) < 2e-3, _rel(gp_f, gp_r)
    for a, b in zip(gd_f, gd_r):
        assert _rel(a, b) < 2e-3, _rel(a, b)
    # planes only (frozen decoder): decoder gradients are not computed and not returned
    dec.requires_grad_(False)
    planes = _t(g['planes'], dev).requires_grad_(True)
    torch.manual_seed(321)
    out = ren(planes, dec, o, d, opts)
    (out[0] * g_rgb).sum().backward()
    assert _rel(planes.grad, _oracle_free_planes_grad(results, g_rgb, ren, dec, g, o, d, opts, dev)) < 2e-3


def test_query_points_backward_vs_oracle(dev):
    """gnerf_query_points_backward against autograd through the fp64 oracle's query_points, and the drop-in run_model
    under autograd against the PyTorch-op path (ragged point count: partial last tile)."""
    import gnerf_hip
    from oracle import render_ref as R
    gen = torch.Generator().manual_seed(9)
    N, P_, hw = 2, 70, (24, 20)
    planes = torch.randn(N, 3, 32, *hw, generator=gen) * 1.5
    dec = R.fold_decoder(torch.randn(64, 32, generator=gen), torch.randn(64, generator=gen) * 0.2,
                         torch.randn(33, 64, generator=gen), torch.randn(33, generator=gen) * 0.2)
    pts = (torch.rand(N, P_, 3, generator=gen) - 0.5) * 1.1          # some points outside the box: zero-padded taps
    g_sigma = torch.randn(N, P_, 1, generator=gen)
    g_rgb = torch.randn(N, P_, 32, generator=gen)
    pl = planes.double().requires_grad_(True)
    dc = [t.double().requires_grad_(True) for t in dec]
    sig, rgb = R.query_points(pl, dc, pts.double(), 1.0)
    ((sig * g_sigma.double()).sum() + (rgb * g_rgb.double()).sum()).backward()
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    gp, gdec = gnerf_hip.query_points_backward(nhwc, N, [t.to(dev) for t in dec], pts.to(dev), 1.0, g_sigma.to(dev), g_rgb.to(dev))
    assert _rel(gnerf_hip.planes_from_nhwc(gp, N).cpu(), pl.grad) < 2e-3
    for a, b in zip(gdec, dc):
        assert _rel(a.cpu(), b.grad) < 2e-3
    # sigma-only gradient, planes only
    gp2, none_dec = gnerf_hip.query_points_backward(nhwc, N, [t.to(dev) for t in dec], pts.to(dev), 1.0, g_sigma.to(dev), None, need_decoder=False)
    assert none_dec is None
    pl2 = planes.double().requires_grad_(True)
    (R.query_points(pl2, [t.double() for t in dec], pts.double(), 1.0)[0] * g_sigma.double()).sum().backward()
    assert _rel(gnerf_hip.planes_from_nhwc(gp2, N).cpu(), pl2.grad) < 2e-3


def test_run_model_training_on_gpu(dev, golden):
    from training.volumetric_rendering.renderer import ImportanceRenderer, sample_from_planes
    from test_host_cpu import Decoder, options_of
    g = golden('render_s12.npz')
    ren = ImportanceRenderer().to(dev)
    dec = Decoder(g).to(dev).requires_grad_(True)
    opts = options_of(g)
    pts = (torch.rand(g['planes'].shape[0], 50, 3, device=dev) - 0.5)
    gs, gc = torch.randn(pts.shape[0], 50, 1, device=dev), torch.randn(pts.shape[0], 50, 32, device=dev)
    res = []
    for fused in (True, False):
        planes = _t(g['planes'], dev).requires_grad_(True)
        dec.zero_grad(set_to_none=True)
        if fused:
            out = ren.run_model(planes, dec, pts, torch.zeros_like(pts), opts)
            assert type(out['sigma'].grad_fn).__name__.startswith('_FusedQuery')
        else:
            out = dec(sample_from_planes(ren.plane_axes.to(dev), planes, pts, padding_mode='zeros', box_warp=opts['box_warp']), None)
        ((out['sigma'] * gs).sum() + (out['rgb'] * gc).sum()).backward()
        res.append((planes.grad.clone(), [p.grad.clone() for p in dec.parameters()]))
    assert _rel(res[0][0], res[1][0]) < 2e-3
    for a, b in zip(res[0][1], res[1][1]):
        assert _rel(a, b) < 2e-3


# ---------------------------------------------------------------------------- ops


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.float64])
def test_bias_act_all_activations_and_orders(dev, dtype):
    from torch_utils.ops import bias_act
    from oracle import ops_ref as O
    torch.manual_seed(0)
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.float64: 2e-7}[dtype]    # fp64: gain/alpha cross the plugin API as float32 (bias_act.cpp:36)
    x0 = (torch.randn(3, 6, 5, 7, dtype=torch.float64) * 2).to(dtype)
    b0 = torch.randn(6, dtype=torch.float64).to(dtype)
    dy0 = torch.randn(3, 6, 5, 7, dtype=torch.float64).to(dtype)
    dd0 = torch.randn(3, 6, 5, 7, dtype=torch.float64).to(dtype)
    xn, bn, dyn, ddn = [t.double().numpy() for t in (x0, b0, dy0, dd0)]
    for act in bias_act.activation_funcs:
        for clamp in (None, 0.875):        # exactly representable in fp16/fp32/fp64 (the plugin takes clamp as a float32 and the
            # backward mask compares the SAVED output with it, so an inexact bound would be dtype-dependent, as in the reference)
            x = x0.to(dev).requires_grad_(True)
            b = b0.to(dev).requires_grad_(True)
            y = bias_act.bias_act(x, b, act=act, clamp=clamp)
            assert y.dtype == dtype
            np.testing.assert_allclose(y.detach().double().cpu().numpy(), O.bias_act(xn, bn, 1, act, clamp=clamp), rtol=tol, atol=tol, err_msg=f'{act} fwd')
            dx, db = torch.autograd.grad(y, (x, b), dy0.to(dev), create_graph=True)
            ref_dx = O.bias_act_grad(dyn, xn, bn, 1, act, clamp=clamp)
            if act == 'linear':
                # Reference GPU-path quirk, kept for drop-in fidelity: 'linear' saves neither x nor y for backward
                # (activation_funcs ref='', bias_act.py:23,155-158), so the plugin's clamp mask sees yref = 0 and
                # passes every gradient (bias_act.cu:137-146) -- unlike autograd of the PyTorch-op path.
                ref_dx = dyn.copy()
            np.testing.assert_allclose(dx.detach().double().cpu().numpy(), ref_dx, rtol=tol, atol=tol, err_msg=f'{act} dx')
            np.testing.assert_allclose(db.detach().double().cpu().numpy(), ref_dx.sum((0, 2, 3)), rtol=tol * 20, atol=tol * 20, err_msg=f'{act} db')
            if bias_act.activation_funcs[act].has_2nd_grad:
                (d2,) = torch.autograd.grad(dx, x, dd0.to(dev))
                np.testing.assert_allclose(d2.double().cpu().numpy(), O.bias_act_grad2(ddn, dyn, xn, bn, 1, act, clamp=clamp), rtol=tol, atol=tol * 4, err_msg=f'{act} d2')


def test_bias_act_layouts_and_edges(dev):
    from torch_utils.ops import bias_act
    from oracle import ops_ref as O
    torch.manual_seed(1)
    x = torch.randn(2, 8, 5, 3, device=dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(8, device=dev)
    y = bias_act.bias_act(x, b, act='lrelu', clamp=256)
    assert y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(y.cpu().numpy(), O.bias_act(x.cpu().numpy(), b.cpu().numpy(), 1, 'lrelu', clamp=256), rtol=1e-6, atol=1e-6)
    x2 = torch.randn(5, 512, device=dev)                       # mapping-network shape, bias on the last dim
    b2 = torch.randn(512, device=dev)
    np.testing.assert_allclose(bias_act.bias_act(x2, b2, act='lrelu').cpu().numpy(), O.bias_act(x2.cpu().numpy(), b2.cpu().numpy(), 1, 'lrelu'), rtol=1e-6, atol=1e-6)
    x3 = torch.randn(1031, device=dev)                          # ragged tail, no bias, dim ignored
    np.testing.assert_allclose(bias_act.bias_act(x3, act='swish').cpu().numpy(), O.bias_act(x3.cpu().numpy(), None, 0, 'swish'), rtol=1e-5, atol=1e-6)
    x4 = torch.randn(4, 3, 9, device=dev)[:, :, 1:]            # non-dense view: the op makes it contiguous first (bias_act.py:147)...
    np.testing.assert_allclose(bias_act.bias_act(x4, act='relu').cpu().numpy(), np.maximum(x4.cpu().numpy(), 0) * np.sqrt(2), rtol=1e-6)
    import gnerf_hip
    with pytest.raises(RuntimeError):                           # ...the native entry point itself rejects it (bias_act.cpp:50)
        gnerf_hip.bias_act(x4, None, None, None, None, 0, 1, 2, 0.0, 1.0, -1.0)
    x5 = torch.randn(40, device=dev)[1:]                       # dense but 4-byte aligned only: scalar kernel
    np.testing.assert_allclose(bias_act.bias_act(x5, act='tanh').cpu().numpy(), np.tanh(x5.cpu().numpy()), rtol=1e-5, atol=1e-6)
    assert bias_act.bias_act(torch.empty(0, 4, device=dev), act='relu').shape == (0, 4)
    big = torch.randn(4, 128, 64, 64, device=dev, dtype=torch.float16)
    bb = torch.randn(128, device=dev, dtype=torch.float16)
    ref = (torch.nn.functional.leaky_relu(big.float() + bb.float()[None, :, None, None], 0.2) * np.sqrt(2)).clamp(-256, 256)
    np.testing.assert_allclose(bias_act.bias_act(big, bb, act='lrelu', clamp=256).float().cpu().numpy(), ref.cpu().numpy(), rtol=2e-3, atol=2e-3)


def test_clamp_sends_nan_to_minus_clamp_like_the_reference_kernel(dev):
    """The forward clamp of bias_act and of the fused epilogues is v_med3_f32: NaN -> -clamp, which is what the reference's CUDA kernel
    computes (bias_act.cu:143: `(y > -clamp & y < clamp) ? y : (y >= 0) ? clamp : -clamp`); +-inf clamp to +-clamp; without a clamp
    NaN passes through.  (The CPU `_ref` path's torch.clamp keeps NaN, bias_act.py:121.)"""
    import gnerf_hip
    from torch_utils.ops import bias_act
    for dtype in (torch.float32, torch.float16):
        x = torch.tensor([float('nan'), float('inf'), -float('inf'), 0.5, -3.0, 300.0], device=dev, dtype=dtype).reshape(1, 6, 1, 1).expand(1, 6, 4, 4).contiguous()
        b = torch.zeros(6, device=dev, dtype=dtype)
        for act in ('linear', 'lrelu'):
            y = bias_act.bias_act(x, b, act=act, gain=1.0, clamp=2.0)[0, :, 0, 0].float().cpu()
            want = torch.tensor([-2.0, 2.0, -2.0, 0.5, -3.0 if act == 'linear' else -0.6, 2.0])
            want = want.clamp(-2.0, 2.0)
            assert torch.allclose(y, want, atol=2e-3), (dtype, act, y)
            y = bias_act.bias_act(x, b, act=act, gain=1.0)[0, :, 0, 0].float().cpu()
            assert torch.isnan(y[0]) and torch.isinf(y[1]) and y[1] > 0
        e = gnerf_hip.modconv_epilogue(x, b, act='lrelu', gain=1.0, clamp=2.0)[0, :, 0, 0].float().cpu()
        assert torch.allclose(e, torch.tensor([-2.0, 2.0, -2.0, 0.5, -0.6, 2.0]), atol=2e-3)
        ec = gnerf_hip.modconv_epilogue(x.contiguous(memory_format=torch.channels_last), b, act='lrelu', gain=1.0, clamp=2.0)
        assert torch.equal(ec.contiguous(), gnerf_hip.modconv_epilogue(x, b, act='lrelu', gain=1.0, clamp=2.0))


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_bias_act_channels_last_kernel(dev, dtype):
    """The channels-last form (bias along the fastest axis, a lane keeps its VEC biases in registers) against the NCHW result of the
    same values, bit for bit: forward for two activations, and the first-order gradient form that reads the saved output."""
    from torch_utils.ops import bias_act
    torch.manual_seed(4)
    for shape in ((1, 128, 40, 24), (3, 32, 17, 9), (2, 264, 5, 7)):          # 264 channels: 33 (fp16) / 66 (fp32) vectors per pixel -> general kernel
        x = torch.randn(*shape, device=dev, dtype=dtype, requires_grad=True)
        b = torch.randn(shape[1], device=dev, dtype=dtype)
        xc = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for act, kw in (('lrelu', dict(clamp=1.5)), ('linear', dict(clamp=0.7)), ('sigmoid', {})):
            ya, yb = bias_act.bias_act(x, b, act=act, **kw), bias_act.bias_act(xc, b, act=act, **kw)
            assert yb.is_contiguous(memory_format=torch.channels_last) and torch.equal(ya, yb), (shape, act)
            g = torch.randn_like(ya)
            ga, = torch.autograd.grad(ya, x, g)
            gb, = torch.autograd.grad(yb, xc, g.contiguous(memory_format=torch.channels_last))
            assert torch.equal(ga, gb), (shape, act)


UP_CASES = {
    'blur':      dict(f='f4', up=1, down=1, padding=[1, 1, 1, 1], gain=4.0),
    'up2':       dict(f='f4', up=2, down=1, padding=[2, 1, 2, 1], gain=4.0),
    'down2':     dict(f='f4', up=1, down=2, padding=[1, 1, 1, 1], gain=1.0),
    'asym':      dict(f='fa', up=[2, 1], down=[1, 2], padding=[1, 2, 3, 0], gain=1.5),
    'asym_flip': dict(f='fa', up=[2, 1], down=[1, 2], padding=[1, 2, 3, 0], gain=1.5, flip_filter=True),
    'crop':      dict(f='f4', up=2, down=1, padding=[-1, 2, 3, -2], gain=1.0),
    'sep':       dict(f='fs', up=2, down=3, padding=[4, 3, 5, 2], gain=2.0),
    'sep_flip':  dict(f='fs', up=1, down=1, padding=[4, 3, 4, 3], gain=1.0, flip_filter=True),
    'none':      dict(f=None, up=2, down=1, padding=0, gain=1.0),
}


@pytest.mark.parametrize('name', list(UP_CASES))
@pytest.mark.parametrize('layout', ['nchw', 'nhwc'])
def test_upfirdn2d_golden(dev, golden, name, layout):
    from torch_utils.ops import upfirdn2d
    g = golden('ops.npz')
    kw = dict(UP_CASES[name])
    f = kw.pop('f')
    f = None if f is None else _t(g['up_' + f], dev)
    x = _t(g['up_x'], dev)
    if layout == 'nhwc':
        x = x.contiguous(memory_format=torch.channels_last)
    y = upfirdn2d.upfirdn2d(x, f, **kw)
    assert tuple(y.shape) == g['up_' + name].shape
    np.testing.assert_allclose(y.cpu().numpy(), g['up_' + name], rtol=1e-5, atol=3e-6)
    if name in ('blur', 'up2', 'down2', 'asym'):
        xg = x.clone().requires_grad_(True)
        (dx,) = torch.autograd.grad(upfirdn2d.upfirdn2d(xg, f, **kw), xg, _t(g[f'up_{name}_dy'], dev))
        np.testing.assert_allclose(dx.cpu().numpy(), g[f'up_{name}_dx'], rtol=1e-5, atol=5e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.float64])
def test_upfirdn2d_shapes_vs_oracle(dev, dtype):
    """StyleGAN2's two hot variants at ragged sizes that straddle tile edges, + helper wrappers."""
    from torch_utils.ops import upfirdn2d
    from oracle import ops_ref as O
    torch.manual_seed(2)
    tol = {torch.float32: 2e-5, torch.float16: 6e-3, torch.float64: 1e-11}[dtype]
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    fn = f.cpu().numpy()
    for shape, kw in [((2, 3, 67, 131), dict(padding=[1, 1, 1, 1], gain=4.0)),
                      ((1, 5, 33, 70), dict(up=2, padding=[2, 1, 2, 1], gain=4.0)),
                      ((2, 2, 40, 150), dict(down=2, padding=[1, 1, 1, 1])),
                      ((1, 2, 5, 3), dict(up=2, padding=[2, 1, 2, 1], gain=4.0))]:
        x = torch.randn(*shape, dtype=torch.float64).to(dtype)
        y = upfirdn2d.upfirdn2d(x.to(dev), f, **kw)
        ref = O.upfirdn2d(x.double().numpy(), fn, **kw)
        assert y.dtype == dtype and tuple(y.shape) == ref.shape
        np.testing.assert_allclose(y.double().cpu().numpy(), ref, rtol=tol, atol=tol)
    x = torch.randn(1, 2, 16, 16, device=dev)
    np.testing.assert_allclose(upfirdn2d.upsample2d(x, f).cpu().numpy(), O.upfirdn2d(x.cpu().numpy(), fn, up=2, padding=[2, 1, 2, 1], gain=4), atol=2e-6)
    np.testing.assert_allclose(upfirdn2d.downsample2d(x, f).cpu().numpy(), O.upfirdn2d(x.cpu().numpy(), fn, down=2, padding=[1, 1, 1, 1]), atol=2e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_upfirdn2d_blur_edge_shapes(dev, dtype):
    """The streaming 4x4 kernel (up = down = 1) at the shapes that stress its edges: images narrower than a lane's 16
    bytes, fewer rows than a block of four, widths that leave the last lane partial, many tiny images per wave (sub-strips),
    asymmetric / negative / large padding, flipped filter, a non-symmetric filter, and the first/last elements of the tensor
    (guarded path)."""
    from torch_utils.ops import upfirdn2d
    from oracle import ops_ref as O
    torch.manual_seed(5)
    tol = {torch.float32: 2e-5, torch.float16: 6e-3}[dtype]
    f_sym = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    f_any = torch.randn(4, 4, generator=torch.Generator().manual_seed(1)).to(dev)
    cases = [((1, 1, 4, 4), [1, 1, 1, 1]), ((1, 1, 3, 3), [2, 1, 2, 1]), ((3, 5, 5, 9), [1, 1, 1, 1]), ((2, 7, 2, 37), [2, 1, 3, 2]),
             ((1, 2, 9, 1), [3, 3, 1, 1]), ((40, 3, 6, 6), [1, 2, 2, 1]), ((1, 1, 70, 1030), [1, 1, 1, 1]), ((2, 2, 33, 257), [0, 3, 3, 0]),
             ((1, 3, 12, 20), [-1, 2, 4, -2]), ((1, 1, 8, 8), [5, 5, 5, 5])]
    for shape, pad in cases:
        for f, flip in ((f_sym, False), (f_any, False), (f_any, True)):
            x = torch.randn(*shape, dtype=torch.float64).to(dtype)
            y = upfirdn2d.upfirdn2d(x.to(dev), f, padding=pad, flip_filter=flip, gain=1.7)
            ref = O.upfirdn2d(x.double().numpy(), f.cpu().numpy(), padding=pad, flip_filter=flip, gain=1.7)
            assert tuple(y.shape) == ref.shape, (shape, pad)
            scale = max(1.0, float(np.abs(ref).max()))
            np.testing.assert_allclose(y.double().cpu().numpy(), ref, rtol=tol, atol=tol * scale, err_msg=str((shape, pad, flip)))


def test_upfirdn2d_linearity_at_full_size(dev):
    """[4,128,513,513] -> 512x512 blur (the largest call of a G-NeRF forward, fp16): too big for the numpy oracle;
    upfirdn is linear, so f(a*x + y) == a*f(x) + f(y), and a constant image maps to the filter's DC gain away from edges."""
    from torch_utils.ops import upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    x = torch.randn(2, 128, 513, 513, device=dev, dtype=torch.float16)
    y = torch.randn_like(x)
    kw = dict(padding=[1, 1, 1, 1], gain=4.0)
    lhs = upfirdn2d.upfirdn2d((0.5 * x.float() + y.float()), f, **kw)
    rhs = 0.5 * upfirdn2d.upfirdn2d(x.float(), f, **kw) + upfirdn2d.upfirdn2d(y.float(), f, **kw)
    assert lhs.shape == (2, 128, 512, 512)
    np.testing.assert_allclose(lhs.cpu().numpy(), rhs.cpu().numpy(), atol=1e-4)
    h = upfirdn2d.upfirdn2d(x, f, **kw)
    np.testing.assert_allclose(h.float().cpu().numpy(), upfirdn2d.upfirdn2d(x.float(), f, **kw).cpu().numpy(), atol=3e-2, rtol=2e-3)
    ones = torch.ones(1, 1, 64, 64, device=dev)
    np.testing.assert_allclose(upfirdn2d.upfirdn2d(ones, f, **kw)[0, 0, 4:-4, 4:-4].cpu().numpy(), 4.0, rtol=1e-6)


def test_filtered_lrelu_gpu(dev, golden):
    from torch_utils.ops import filtered_lrelu
    g = golden('ops.npz')
    x, b, fu, fd = _t(g['fl_x'], dev), _t(g['fl_b'], dev), _t(g['fl_fu'], dev), _t(g['fl_fd'], dev)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        y = filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=2, down=2, padding=[10, 10, 10, 10], gain=1.3, slope=0.1, clamp=0.8)
        np.testing.assert_allclose(y.cpu().numpy(), g['fl_up2_down2'], rtol=1e-4, atol=3e-6)
        y = filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=4, down=2, padding=[11, 10, 9, 12], flip_filter=True)
        np.testing.assert_allclose(y.cpu().numpy(), g['fl_up4_down2'], rtol=1e-4, atol=3e-6)
        np.testing.assert_allclose(filtered_lrelu.filtered_lrelu(x, b=b).cpu().numpy(), g['fl_plain'], rtol=1e-5, atol=1e-6)
        # gradient through the sign tensor == autograd of the PyTorch-op path
        xg = x.clone().requires_grad_(True)
        bg = b.clone().requires_grad_(True)
        yy = filtered_lrelu.filtered_lrelu(xg, fu=fu, fd=fd, b=bg, up=2, down=2, padding=[10, 10, 10, 10], gain=1.3, slope=0.1, clamp=0.8)
        gy = torch.randn_like(yy)
        dx, db = torch.autograd.grad(yy, (xg, bg), gy)
        xr = x.detach().cpu().clone().requires_grad_(True)
        br = b.detach().cpu().clone().requires_grad_(True)
        yr = filtered_lrelu.filtered_lrelu(xr, fu=fu.cpu(), fd=fd.cpu(), b=br, up=2, down=2, padding=[10, 10, 10, 10], gain=1.3, slope=0.1, clamp=0.8, impl='ref')
        dxr, dbr = torch.autograd.grad(yr, (xr, br), gy.cpu())
    np.testing.assert_allclose(dx.cpu().numpy(), dxr.numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(db.cpu().numpy(), dbr.numpy(), rtol=1e-4, atol=5e-5)


# ---- fused filtered_lrelu kernel (csrc/filtered_lrelu_fused.hip) -----------------------------------------------------

def _fl_filter(taps, seed):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal(taps).astype(np.float32)
    return f / np.abs(f).sum()


_FL_CASES = [
    # up, fu taps, down, fd taps, padding, flip, shape
    (2, 12, 2, 12, [10, 10, 10, 10], False, (2, 3, 20, 24)),
    (2, 12, 2, 12, [11, 9, 12, 8], True, (1, 2, 37, 41)),          # several tiles, odd phases
    (4, 24, 2, 12, [20, 21, 19, 22], False, (1, 2, 18, 22)),
    (2, 12, 4, 24, [15, 16, 17, 14], True, (1, 2, 40, 36)),
    (4, 32, 4, 32, [31, 30, 29, 32], False, (1, 1, 24, 20)),
    (2, 16, 2, 16, [12, 13, 14, 11], False, (1, 3, 19, 17)),
    (2, 8, 1, 1, [4, 3, 5, 2], False, (2, 2, 15, 70)),            # no downsampling filter
    (1, 1, 2, 8, [3, 4, 2, 5], True, (2, 2, 33, 31)),             # no upsampling filter
    (1, 1, 1, 1, [0, 0, 0, 0], False, (2, 5, 9, 13)),             # bias + activation only
    (2, 3, 2, 5, [3, 2, 3, 2], False, (1, 2, 16, 16)),            # short filters, zero-padded to the branch size
    (4, 24, 4, 24, [22, 23, 21, 24], True, (1, 1, 70, 66)),
]


@pytest.mark.parametrize('case', _FL_CASES, ids=lambda c: f'up{c[0]}x{c[1]}_down{c[2]}x{c[3]}')
@pytest.mark.parametrize('dtype', ['float32', 'float16'])
def test_filtered_lrelu_fused_vs_oracle(dev, case, dtype):
    """The single-launch kernel through the plugin entry point (return code 0 = it ran fused) against the numpy oracle."""
    import gnerf_hip
    from oracle import ops_ref as O
    up, fut, down, fdt, pad, flip, shape = case
    rng = np.random.default_rng(7)
    x = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape[1]).astype(np.float32)
    fu = _fl_filter(fut, 1) if fut > 1 else None
    fd = _fl_filter(fdt, 2) if fdt > 1 else None
    gain, slope, clamp = 1.7, 0.15, 0.6
    td = getattr(torch, dtype)
    xt, bt = _t(x, dev).to(td), _t(b, dev).to(td)
    fut_t = _t(fu, dev) if fu is not None else torch.ones([1, 1], device=dev)
    fdt_t = _t(fd, dev) if fd is not None else torch.ones([1, 1], device=dev)
    y, so, rc = gnerf_hip.filtered_lrelu(xt, fut_t, fdt_t, bt, torch.empty([0]), up, down, *pad, 0, 0, gain, slope, clamp, flip, True)
    assert rc == 0, 'configuration must be covered by the fused kernel'
    ref = O.filtered_lrelu(xt.float().cpu().numpy(), fu, fd, bt.float().cpu().numpy(), up=up, down=down, padding=pad,
                           gain=gain, slope=slope, clamp=clamp, flip_filter=flip)
    assert tuple(y.shape) == ref.shape
    if dtype == 'float32':
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-4, atol=2e-5)
    else:
        np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=4e-3, atol=2e-3)
    # sign tensor: recompute the activated intermediate with the oracle and compare the two bits per pixel where the
    # value is not within rounding of a decision boundary
    if dtype == 'float32':
        u = O.upfirdn2d(O.bias_act(xt.float().cpu().numpy(), bt.float().cpu().numpy()), fu, up=up, padding=pad, gain=up ** 2, flip_filter=flip) * gain
        s_h, s_w = so.shape[2], so.shape[3] * 4
        hh, ww = min(u.shape[2], s_h), min(u.shape[3], (y.shape[3] - 1) * down + 1 + (fdt - 1))   # active width (filtered_lrelu.cpp:96)
        sb = so.cpu().numpy()
        bits = np.stack([(sb >> (2 * k)) & 3 for k in range(4)], axis=-1).reshape(*sb.shape[:3], -1)[:, :, :hh, :ww]
        uu = u[:, :, :hh, :ww]
        lr = np.where(uu < 0, uu * slope, uu)
        want = np.where(np.abs(lr) > clamp, 2, (uu < 0).astype(np.int64))
        safe = (np.abs(uu) > 1e-5) & (np.abs(np.abs(lr) - clamp) > 1e-5)
        assert hh >= (y.shape[2] - 1) * down + 1 and safe.mean() > 0.25     # (zero padding leaves exact zeros: not 'safe')
        bad = np.argwhere((bits != want) & safe)
        assert len(bad) == 0, (len(bad), bad[:8].tolist(), [(float(uu[tuple(i)]), int(bits[tuple(i)]), int(want[tuple(i)])) for i in bad[:8]])


@pytest.mark.parametrize('case', [_FL_CASES[1], _FL_CASES[2], _FL_CASES[3], _FL_CASES[6], _FL_CASES[7]],
                         ids=lambda c: f'up{c[0]}x{c[1]}_down{c[2]}x{c[3]}')
def test_filtered_lrelu_fused_gradient(dev, case):
    """Backward = the fused kernel again with up/down swapped, reading the sign tensor (filtered_lrelu.py:254-265),
    against autograd through the PyTorch-op form on the CPU."""
    from torch_utils.ops import filtered_lrelu
    import gnerf_hip
    up, fut, down, fdt, pad, flip, shape = case
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(shape[1]).astype(np.float32))
    fu = torch.from_numpy(_fl_filter(fut, 1)) if fut > 1 else None
    fd = torch.from_numpy(_fl_filter(fdt, 2)) if fdt > 1 else None
    kw = dict(up=up, down=down, padding=pad, gain=1.7, slope=0.15, clamp=0.6, flip_filter=flip)
    calls = []
    real = gnerf_hip.filtered_lrelu
    plugin = filtered_lrelu.custom_ops.get_plugin('filtered_lrelu_plugin', sources=[])

    def spy(*a):
        r = real(*a)
        calls.append(r[2])
        return r
    plugin.filtered_lrelu = staticmethod(spy)
    try:
        xg, bg = x.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        y = filtered_lrelu.filtered_lrelu(xg, fu=None if fu is None else fu.to(dev), fd=None if fd is None else fd.to(dev), b=bg, **kw)
        gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
        dx, db = torch.autograd.grad(y, (xg, bg), gy.to(dev))
    finally:
        plugin.filtered_lrelu = staticmethod(real)
    assert calls == [0, 0], f'forward and backward must both run fused, got return codes {calls}'
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = filtered_lrelu.filtered_lrelu(xr, fu=fu, fd=fd, b=br, impl='ref', **kw)
    dxr, dbr = torch.autograd.grad(yr, (xr, br), gy)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dx.cpu().numpy(), dxr.numpy(), rtol=2e-4, atol=3e-5)
    np.testing.assert_allclose(db.cpu().numpy(), dbr.numpy(), rtol=2e-4, atol=2e-4)


def test_filtered_lrelu_fused_declines_what_it_does_not_cover(dev):
    import gnerf_hip
    from oracle import ops_ref as O
    x = torch.randn(1, 2, 16, 16, device=dev)
    b = torch.zeros(2, device=dev)
    one = torch.ones([1, 1], device=dev)
    f2d = torch.ones([4, 4], device=dev) / 16
    f3 = torch.ones([9], device=dev) / 9
    big = torch.ones([40], device=dev) / 40
    none = torch.empty([0])
    assert gnerf_hip.filtered_lrelu(x, f2d, one, b, none, 2, 1, 2, 1, 2, 1, 0, 0, 1.0, 0.2, 1e9, False, False)[2] == -1     # non-separable
    assert gnerf_hip.filtered_lrelu(x, f3, f3, b, none, 3, 3, 4, 4, 4, 4, 0, 0, 1.0, 0.2, 1e9, False, False)[2] == -1        # factor 3
    assert gnerf_hip.filtered_lrelu(x, big, one, b, none, 2, 1, 20, 19, 20, 19, 0, 0, 1.0, 0.2, 1e9, False, False)[2] == -1  # 20 taps per branch
    assert gnerf_hip.filtered_lrelu(x.double(), one, one, b.double(), none, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0.2, 1e9, False, False)[2] == -1
    # ... and the public op still answers through the three-launch route
    from torch_utils.ops import filtered_lrelu
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        y = filtered_lrelu.filtered_lrelu(x, fu=f2d, up=2, padding=[2, 1, 2, 1])
    ref = O.filtered_lrelu(x.cpu().numpy(), f2d.cpu().numpy(), None, None, up=2, padding=[2, 1, 2, 1])
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


def test_cpp_extension_and_ctypes_bindings_agree(dev):
    """The two bindings of the C ABI -- gnerf_torch_ext (pybind, the default of the public ops) and ctypes -- reach the same
    kernels: bit-identical results for bias_act (forward and both gradient orders), upfirdn2d (three variants, NCHW and
    channels_last), filtered_lrelu (+ signs, + the -1 return code) and the fused renderer."""
    import gnerf_hip
    e = gnerf_hip.ext()
    assert e is not None, 'gnerf_torch_ext.so missing on the GPU box'
    null = torch.empty([0])
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 9, 11, generator=gen).to(dev)
    b = torch.randn(6, generator=gen).to(dev)
    dy = torch.randn(2, 6, 9, 11, generator=gen).to(dev)
    for dt in (torch.float32, torch.float16):
        xx, bb, dd = x.to(dt), b.to(dt), dy.to(dt)
        y1, y2 = e.bias_act(xx, bb, null, null, null, 0, 1, 3, 0.2, 1.41, 0.5), gnerf_hip.bias_act(xx, bb, None, None, None, 0, 1, 3, 0.2, 1.41, 0.5)
        assert torch.equal(y1, y2)
        g1, g2 = e.bias_act(dd, bb, null, y1, null, 1, 1, 3, 0.2, 1.41, 0.5), gnerf_hip.bias_act(dd, bb, None, y2, None, 1, 1, 3, 0.2, 1.41, 0.5)
        assert torch.equal(g1, g2)
        cl = xx.contiguous(memory_format=torch.channels_last)
        assert torch.equal(e.bias_act(cl, bb, null, null, null, 0, 1, 5, 0.0, 1.0, -1.0), gnerf_hip.bias_act(cl, bb, None, None, None, 0, 1, 5, 0.0, 1.0, -1.0))
        f = torch.tensor([[1., 3, 3, 1]], device=dev).t() @ torch.tensor([[1., 3, 3, 1]], device=dev) / 64
        for args in ((1, 1, 1, 1, 1, 1, 1, 1, False, 4.0), (2, 2, 1, 1, 2, 1, 2, 1, False, 4.0), (1, 1, 2, 2, 1, 1, 1, 1, True, 1.0)):
            for t in (xx, cl):
                u1, u2 = e.upfirdn2d(t, f, *args), gnerf_hip.upfirdn2d(t, f, *args)
                assert u1.stride() == u2.stride() and torch.equal(u1, u2)
    f1 = torch.tensor([1., 3, 3, 1], device=dev) / 8
    for write in (False, True):
        a = e.filtered_lrelu(x, f1, f1, b, null, 2, 2, 3, 2, 3, 2, 0, 0, 1.41, 0.2, 0.8, False, write)
        c = gnerf_hip.filtered_lrelu(x, f1, f1, b, None, 2, 2, 3, 2, 3, 2, 0, 0, 1.41, 0.2, 0.8, False, write)
        assert a[2] == c[2] == 0 and torch.equal(a[0], c[0]) and a[1].shape == c[1].shape
        if write:                       # bytes past the valid width (the row pitch is rounded up to 16 pixels) are never written
            valid = (a[0].shape[3] * 2 - 1 + 3) // 4
            assert torch.equal(a[1][..., :valid], c[1][..., :valid])
    assert e.filtered_lrelu(x, f1, f1, b, null, 3, 3, 3, 2, 3, 2, 0, 0, 1.0, 0.2, -1.0, False, False)[2] == -1
    xa, xb = x.clone(), x.clone()
    assert torch.equal(e.filtered_lrelu_act_(xa, null, 0, 0, 1.41, 0.2, 0.5, True), gnerf_hip.filtered_lrelu_act_(xb, None, 0, 0, 1.41, 0.2, 0.5, True)) and torch.equal(xa, xb)
    # renderer: the default route (extension) against the ctypes route forced through the debug flag's code path
    planes, dec, o, d, nc, nf = _random_scene(3, N=2, res=8, S=48, F=48, hw=(16, 16))
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    args = (nhwc, 2, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev))
    kw = dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=8)
    via_ext = gnerf_hip.render_forward(*args, **kw)
    via_ctypes = gnerf_hip.render_forward(*args, debug=True, **kw)[:3]
    for p_, q_ in zip(via_ext, via_ctypes):
        assert torch.equal(p_, q_)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_modconv_kernels_vs_composed_ops(dev, dtype):
    """csrc/modconv.hip against the PyTorch-op chains of modulated_conv2d / SynthesisLayer.forward they replace
    (networks_stylegan2.py:61-86, :315-334): modulated + demodulated weights with the fp16 pre-normalisation, the demodulation
    coefficients alone, the input scaling, and the fused demodulation + noise + bias + lrelu + gain + clamp epilogue in both
    the fused-convolution (x.add_(noise)) and un-fused (fma) orders."""
    import gnerf_hip
    import gnerf_generator as GG
    from torch_utils.ops import bias_act
    gen = torch.Generator().manual_seed(0)
    half = dtype == torch.float16
    for (n, o, i, k) in [(4, 64, 32, 3), (1, 96, 256, 1), (3, 8, 5, 3)]:
        weight = torch.randn(o, i, k, k, generator=gen).to(dev)
        styles = (torch.randn(n, i, generator=gen) + 1).to(dev)
        for demod in (True, False):
            ref = GG._modulated_weights(weight, styles, demod, half).to(dtype)
            got, dco = gnerf_hip.modulate_weights(weight, styles, demod, out_dtype=dtype, want_dcoefs=True)
            assert got.shape == (n, o, i, k, k) and got.dtype == dtype
            # the other memory orders hold the same values: [N,I,O,k,k] for conv_transpose2d, channels_last memory of either
            for tr in (False, True):
                for cl in (False, True):
                    alt, _ = gnerf_hip.modulate_weights(weight, styles, demod, out_dtype=dtype, transposed=tr, channels_last=cl)
                    assert torch.equal(alt, got.transpose(1, 2) if tr else got)
                    assert alt[0].is_contiguous(memory_format=torch.channels_last) if cl else alt.is_contiguous()
            np.testing.assert_allclose(got.float().cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-3 if half else 3e-6, atol=1e-7)
            if demod:
                w_, s_ = GG._prenormalize(weight, styles) if half else (weight, styles)
                np.testing.assert_allclose(dco.cpu().numpy(), GG._demod_coefficients(w_, s_).cpu().numpy(), rtol=3e-6)
                if half:
                    np.testing.assert_allclose(gnerf_hip.normalise_styles(styles).cpu().numpy(), s_.cpu().numpy(), rtol=1e-6)
            else:
                assert dco is None
    x = (torch.randn(3, 8, 12, 16, generator=gen) * 3).to(dev).to(dtype)
    sc = (torch.randn(3, 8, generator=gen) + 1).to(dev)
    assert torch.equal(gnerf_hip.scale_channels(x, sc), x * sc.to(dtype)[:, :, None, None])
    b = torch.randn(8, generator=gen).to(dev)
    tol = dict(rtol=2e-3, atol=2e-3) if half else dict(rtol=2e-6, atol=2e-6)
    for noise in (None, torch.randn(12, 16, generator=gen).to(dev), torch.randn(3, 1, 12, 16, generator=gen).to(dev)):
        for clamp in (None, 1.5):
            # fused-convolution order: x.add_(noise) in x's dtype, then bias_act
            want = bias_act.bias_act(x.clone().add_(noise) if noise is not None else x, b.to(dtype), act='lrelu', gain=1.3, clamp=clamp)
            got = gnerf_hip.modconv_epilogue(x, b, noise=noise, act='lrelu', gain=1.3, clamp=clamp)
            np.testing.assert_allclose(got.float().cpu().numpy(), want.float().cpu().numpy(), **tol)
            # un-fused order: fma(x, dcoefs, noise) in x's dtype, then bias_act
            d4 = sc.to(dtype)[:, :, None, None]
            want = bias_act.bias_act(torch.addcmul(noise.to(dtype), x, d4) if noise is not None else x * d4, b.to(dtype), act='lrelu', gain=1.3, clamp=clamp)
            got = gnerf_hip.modconv_epilogue(x, b, scale=sc, noise=noise, round_noise=True, act='lrelu', gain=1.3, clamp=clamp)
            np.testing.assert_allclose(got.float().cpu().numpy(), want.float().cpu().numpy(), **tol)
    want = bias_act.bias_act(x, b.to(dtype), clamp=2.0)
    assert torch.equal(gnerf_hip.modconv_epilogue(x, b, act='linear', clamp=2.0), want)
    with pytest.raises(RuntimeError):
        gnerf_hip.modconv_epilogue(x, b, act='relu')


def test_conv_transpose3x3_s2_vs_framework(dev):
    """csrc/conv3x3.hip MODE 1: the stride-2 transposed 3x3 convolution of the x2 layers as four phase convolutions on the matrix cores,
    against conv_transpose2d (MIOpen) and against the same product in fp32 on the same fp16 operands: odd sizes (the 8 x 32 position tiles
    overhang), several images, several output-channel groups, one to four input chunks."""
    import gnerf_hip
    import torch.nn.functional as F
    for (n, cin, cout, h, w) in [(1, 64, 128, 8, 32), (2, 128, 128, 5, 7), (1, 64, 256, 16, 40), (3, 192, 128, 9, 33), (1, 256, 128, 2, 1), (2, 32, 256, 6, 9), (1, 72, 128, 3, 35)]:
        g = torch.Generator(device='cpu').manual_seed(n + h)
        x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (2 * cin ** 0.5)).to(dev)
        assert gnerf_hip.conv_transpose3x3_s2_supported(x, cout)
        got = gnerf_hip.conv_transpose3x3_s2(x, gnerf_hip.pack_conv_transpose3x3_weights(wt))
        want = F.conv_transpose2d(x, wt.half().transpose(0, 1).contiguous(memory_format=torch.channels_last), stride=2)
        ref = F.conv_transpose2d(x.float(), wt.half().float().transpose(0, 1), stride=2)
        assert got.shape == (n, cout, 2 * h + 1, 2 * w + 1) and got.dtype == torch.float16 and gnerf_hip.is_channels_last(got)
        top = float(ref.abs().max())
        e_got, e_want = float((got.float() - ref).abs().max()), float((want.float() - ref).abs().max())
        assert e_got <= max(1.5 * e_want, 2e-3 * top), (n, cin, cout, h, w, e_got, e_want, top)
    # the HOT shape at full size (round 6): block1.conv0 of the superresolution at a batch of four, 256 -> 128 from 256^2 to 513^2 -- every
    # 64th output pixel against the fp32 product of the same fp16 operands (the whole tensor is 2 x 135 M values; the rest must be finite)
    g = torch.Generator(device='cpu').manual_seed(11)
    x = (torch.randn(4, 256, 256, 256, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(128, 256, 3, 3, generator=g) / (2 * 256 ** 0.5)).to(dev)
    got = gnerf_hip.conv_transpose3x3_s2(x, gnerf_hip.pack_conv_transpose3x3_weights(wt))
    assert got.shape == (4, 128, 513, 513) and bool(torch.isfinite(got).all())
    ref = F.conv_transpose2d(x.float(), wt.half().float().transpose(0, 1), stride=2)[:, :, 3::8, 5::8]
    want = F.conv_transpose2d(x, wt.half().transpose(0, 1).contiguous(memory_format=torch.channels_last), stride=2)[:, :, 3::8, 5::8]
    e_got, e_want = float((got[:, :, 3::8, 5::8].float() - ref).abs().max()), float((want.float() - ref).abs().max())
    assert e_got <= max(1.5 * e_want, 2e-3 * float(ref.abs().max())), (e_got, e_want)
    del x, got, ref, want
    bad = torch.zeros(1, 64, 8, 32, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    assert not gnerf_hip.conv_transpose3x3_s2_supported(bad, 64)                          # output channels come in blocks of 128
    with pytest.raises(RuntimeError):
        gnerf_hip.conv_transpose3x3_s2(bad, torch.zeros(9, 64, 64, device=dev, dtype=torch.float16))


def test_conv_transpose_job_shapes_agree(dev, monkeypatch):
    """The transposed convolution's launcher picks a job shape per call (whole position tiles / phase pairs / single phases per workgroup,
    csrc/conv3x3.hip launch_conv_transpose): the three give the SAME bits -- a phase's sum is taken in one order whoever computes it -- on a
    launch above the chip's 512 workgroup slots (odd sizes: the tile grid overhangs the odd phases), in the fp16 and the fp32-grade form, and
    the launcher's own choice is one of them."""
    import gnerf_hip
    import torch.nn.functional as F
    for (n, cin, cout, h, w) in [(8, 64, 128, 127, 129), (10, 136, 256, 70, 65)]:
        g = torch.Generator(device='cpu').manual_seed(h)
        x32 = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).contiguous(memory_format=torch.channels_last)
        x = x32.half()
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (2 * cin ** 0.5)).to(dev)
        wp = gnerf_hip.pack_conv_transpose3x3_weights(wt)
        assert n * ((h + 8) // 8) * ((w + 32) // 32) * (cout // 128) > 512
        monkeypatch.delenv('GNERF_CONVT_PHASE_JOBS', raising=False)
        own = gnerf_hip.conv_transpose3x3_s2(x, wp)
        outs = []
        for mode in ('0', '1', '2'):
            monkeypatch.setenv('GNERF_CONVT_PHASE_JOBS', mode)
            outs.append(gnerf_hip.conv_transpose3x3_s2(x, wp))
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[0], own), (n, cin, cout, h, w)
        ref = F.conv_transpose2d(x.float(), wt.half().float().transpose(0, 1), stride=2)
        assert float((own.float() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
        if cin % 8 == 0 and gnerf_hip.conv_transpose3x3_s2_f32x3_supported(x32, cout):
            w3 = gnerf_hip.pack_conv_transpose3x3_weights_f32x3(wt)
            x3 = gnerf_hip.split_f16x3(x32)
            o32 = []
            for mode in ('0', '1', '2'):
                monkeypatch.setenv('GNERF_CONVT_PHASE_JOBS', mode)
                o32.append(gnerf_hip.conv_transpose3x3_s2_f32x3(x3, w3))
            assert torch.equal(o32[0], o32[1]) and torch.equal(o32[0], o32[2])
            ref64 = F.conv_transpose2d(x32.double(), wt.double().transpose(0, 1), stride=2)
            assert float((o32[0].double() - ref64).abs().max()) <= 2e-5 * float(ref64.abs().max())
        monkeypatch.delenv('GNERF_CONVT_PHASE_JOBS', raising=False)


def test_conv3x3_epilogue_torgb_vs_two_launches(dev):
    """gnerf_conv3x3_epilogue_torgb_nhwc (ABI 11) -- a block's last layer with the block's ToRGB in its epilogue, the result added to the running
    image, no layer output stored (networks_stylegan2.py:452-463 for the superresolution's final block) -- against the two launches it replaces,
    conv3x3_epilogue then torgb_channels_last(accumulate_into=img): the same roundings (the layer's result rounded to fp16, f16 weights x styles,
    v_dot2_f32_f16, the sum rounded to fp16, bias + clamp, rounded again), so the images agree to the order of an fp32 sum in front of an fp16
    rounding; with and without noise / ToRGB bias / clamps, several images and input-channel chunks, tiles at every border."""
    import gnerf_hip
    gen = torch.Generator().manual_seed(5)
    for (n, cin, h, w, with_noise, with_rgb_bias, clamps) in [(2, 64, 16, 64, False, True, True), (1, 128, 8, 32, True, False, False), (3, 256, 24, 32, False, True, False),
                                                               (2, 72, 8, 96, True, True, True)]:
        x = (torch.randn(n, cin, h, w, generator=gen) * 0.7).to(dev).half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(128, cin, 3, 3, generator=gen) / (3 * cin ** 0.5)).to(dev)
        wpk = gnerf_hip.pack_conv3x3_weights(wt)
        dco = (torch.rand(n, 128, generator=gen) + 0.5).to(dev)
        bias = (torch.randn(128, generator=gen) * 0.1).to(dev)
        noise = (torch.randn(h, w, generator=gen) * 0.05).to(dev) if with_noise else None
        rgb_weight = torch.randn(3, 128, 1, 1, generator=gen).to(dev)
        rgb_styles = ((torch.rand(n, 128, generator=gen) + 0.5) / 128 ** 0.5).to(dev)
        rgb_bias = (torch.randn(3, generator=gen) * 0.2).to(dev) if with_rgb_bias else None
        img0 = torch.randn(n, 3, h, w, generator=gen).to(dev)
        kw = dict(bias=bias, scale=dco, noise=noise, round_noise=True, gain=2 ** 0.5, clamp=256.0 if clamps else None)
        y = gnerf_hip.conv3x3_epilogue(x, wpk, **kw)
        want = gnerf_hip.torgb_channels_last(y, rgb_weight, rgb_styles, rgb_bias, clamp=0.4 if clamps else None, accumulate_into=img0.clone())
        assert gnerf_hip.conv3x3_epilogue_torgb_supported(x, 128)
        got = gnerf_hip.conv3x3_epilogue_torgb(x, wpk, img0.clone(), gnerf_hip.torgb_weights(rgb_weight, rgb_styles), rgb_bias, 0.4 if clamps else None, **kw)
        delta_want = (want - img0)
        top = float(delta_want.abs().max())
        diff = (got - want).abs()
        assert top > 0.05 and float(diff.max()) <= 2 ** -9 * max(top, 1.0), (n, cin, h, w, float(diff.max()), top)       # an fp16 ulp of the layer's output at most
        assert float((diff > 0).float().mean()) < 0.05, float((diff > 0).float().mean())
        if clamps:
            assert float(delta_want.abs().max()) <= 0.4 + 1e-3
    assert not gnerf_hip.conv3x3_epilogue_torgb_supported(x, 256)
    with pytest.raises(RuntimeError):                           # a demodulation scale is part of the form
        gnerf_hip.conv3x3_epilogue_torgb(x, wpk, img0.clone(), gnerf_hip.torgb_weights(rgb_weight, rgb_styles), scale=None)


def test_conv3x3_epilogue_vs_composed_ops(dev):
    """csrc/conv3x3.hip -- the 3x3 convolution of a modulated-convolution layer and its epilogue in one launch (SURVEY 8(f)3;
    networks_stylegan2.py:41-98 as SynthesisLayer.forward calls it, :315-334) -- against the two launches it replaces: torch's
    conv2d on the same fp16 channels_last tensors followed by gnerf_hip.modconv_epilogue, for every combination of demodulation scale,
    noise and next-layer scale; tiles at every image border (zero padding), two channel chunks, two output-channel groups.  Both are
    fp16 results of an fp32-accumulated convolution: they may differ by the rounding of a sum taken in another order (one fp16 ulp on
    few elements), and each is held to an fp32 evaluation of the whole chain."""
    import gnerf_hip
    from torch_utils.ops import bias_act
    gen = torch.Generator().manual_seed(2)
    for (n, cin, cout, h, w) in [(2, 128, 128, 16, 64), (1, 256, 128, 8, 32), (3, 128, 256, 24, 32), (2, 64, 128, 8, 64), (1, 32, 128, 8, 32), (1, 104, 128, 8, 32)]:
        x = (torch.randn(n, cin, h, w, generator=gen) * 0.7).to(dev).half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 3, 3, generator=gen) / (3 * cin ** 0.5)).to(dev)
        w16 = wt.half().contiguous(memory_format=torch.channels_last)
        wpk = gnerf_hip.pack_conv3x3_weights(wt)
        assert wpk.shape == (9, cout, -(-cin // 64) * 64) and torch.equal(wpk[5][:, :cin], wt.half()[:, :, 1, 2]) and not wpk[:, :, cin:].any()
        assert gnerf_hip.conv3x3_epilogue_supported(x, cout)
        sc, nx = (torch.rand(n, cout, generator=gen) + 0.5).to(dev), (torch.rand(n, cout, generator=gen) + 0.5).to(dev)
        bias = (torch.randn(cout, generator=gen) * 0.2).to(dev)
        noise = (torch.randn(h, w, generator=gen) * 0.1).to(dev)
        y32 = torch.nn.functional.conv2d(x.float(), w16.float(), padding=1)
        for scale in (None, sc):
            for nz in (None, noise):
                for nxt in (None, nx):
                    for clamp in (None, 0.75):
                        kw = dict(scale=scale, noise=nz, round_noise=True, gain=1.3, clamp=clamp, next_scale=nxt)
                        got = gnerf_hip.conv3x3_epilogue(x, wpk, bias, **kw)
                        assert got.shape == (n, cout, h, w) and got.dtype == torch.float16 and gnerf_hip.is_channels_last(got)
                        want = gnerf_hip.modconv_epilogue(torch.nn.functional.conv2d(x, w16, padding=1), bias, act='lrelu', **kw)
                        # the chain in fp32 on the same fp16 operands
                        t = y32 * (scale[:, :, None, None] if scale is not None else 1.0) + (nz if nz is not None else 0.0)
                        ref = bias_act.bias_act(t, bias, act='lrelu', gain=1.3, clamp=clamp) * (nxt[:, :, None, None] if nxt is not None else 1.0)
                        top = float(ref.abs().max())
                        e_got, e_want = float((got.float() - ref).abs().max()), float((want.float() - ref).abs().max())
                        # round 6: the fused epilogue runs in fp32 on the accumulators and rounds ONCE (the output); the two-launch form rounds
                        # the convolution, the demodulated value and the activated value to fp16 on the way.  The bar is the fp32 chain: the
                        # fused result may be no further from it than the two-launch form is, and no further than one output rounding (half an
                        # fp16 ulp of the largest value = 2^-11 top, with the convolution's own summation order on top).  Bit-equality with the
                        # two-launch form is NOT required (it was the round-5 criterion); the two only have to agree to the latter's error.
                        assert e_got <= e_want and e_got <= 1.2e-3 * top, (n, cin, cout, scale is not None, nz is not None, nxt is not None, clamp, e_got, e_want, top)
                        assert float((got.float() - want.float()).abs().max()) <= 6e-3 * top
    # the HOT shape at full size (round 6): block1.conv1 of the superresolution at a batch of four, 128 -> 128 @ 512^2, with everything the
    # layer's epilogue does -- every 64th pixel against the fp32 chain on the same fp16 operands, the rest finite
    x = (torch.randn(4, 128, 512, 512, generator=gen) * 0.7).to(dev).half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(128, 128, 3, 3, generator=gen) / (3 * 128 ** 0.5)).to(dev)
    w16 = wt.half().contiguous(memory_format=torch.channels_last)
    sc, nx = (torch.rand(4, 128, generator=gen) + 0.5).to(dev), (torch.rand(4, 128, generator=gen) + 0.5).to(dev)
    bias, noise = (torch.randn(128, generator=gen) * 0.2).to(dev), (torch.randn(512, 512, generator=gen) * 0.1).to(dev)
    kw = dict(scale=sc, noise=noise, round_noise=True, gain=2 ** 0.5, clamp=256.0, next_scale=nx)
    got = gnerf_hip.conv3x3_epilogue(x, gnerf_hip.pack_conv3x3_weights(wt), bias, **kw)
    assert got.shape == (4, 128, 512, 512) and bool(torch.isfinite(got).all())
    y32 = torch.nn.functional.conv2d(x.float(), w16.float(), padding=1)[:, :, 3::8, 5::8]
    ref = bias_act.bias_act(y32 * sc[:, :, None, None] + noise[3::8, 5::8], bias, act='lrelu', gain=2 ** 0.5, clamp=256.0) * nx[:, :, None, None]
    want = gnerf_hip.modconv_epilogue(torch.nn.functional.conv2d(x, w16, padding=1), bias, act='lrelu', **kw)[:, :, 3::8, 5::8]
    e_got, e_want = float((got[:, :, 3::8, 5::8].float() - ref).abs().max()), float((want.float() - ref).abs().max())
    assert e_got <= e_want and e_got <= 1.2e-3 * float(ref.abs().max()), (e_got, e_want)
    del x, got, y32, ref, want
    # shapes the kernel does not tile are refused, not approximated
    bad = torch.zeros(1, 64, 8, 24, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    assert not gnerf_hip.conv3x3_epilogue_supported(bad, 128)                             # widths come in tiles of 32
    with pytest.raises(RuntimeError):
        gnerf_hip.conv3x3_epilogue(bad, torch.zeros(9, 128, 64, device=dev, dtype=torch.float16))


def test_conv_f32x3_vs_float64(dev):
    """Round 6: the fp32-GRADE convolutions for the backbone's float32 layers (csrc/conv3x3.hip OUT32; networks_stylegan2.py:41-98 and
    conv2d_resample.py:109-131 on float32 activations).  gnerf_split_f16x3_nhwc + gnerf_conv3x3_f32x3_epilogue_nhwc / the transposed form against
    a float64 evaluation of the same chain, next to the framework's own float32 convolution: the three-f16-products form must be float32-
    grade.  Its error has two parts: the splits (dropped lo * lo term, roundings of the lo halves: ~2^-21 relative per product, random) and the
    ACCUMULATION, which on the matrix instruction is a float32 chain over the products one after the other (tools/probes/mfma_rounding_probe.hip)
    -- sqrt(K) 2^-24 |partial sum| for K = 9 cin terms, where a blocked float32 summation (what MIOpen's kernel amounts to) has 2^-24 sqrt(K / 32).
    A numpy model of the two orders on K = 4608 gives 1.1e-3 against 1.5e-4 at |sum| <= 470: the test's bound is the chain's, and no more than
    four times the framework's where that is larger."""
    import gnerf_hip
    from torch_utils.ops import bias_act
    gen = torch.Generator().manual_seed(5)
    gnerf_hip.split_overflow_flag(dev).zero_()
    for (n, cin, cout, h, w) in [(2, 128, 128, 16, 64), (1, 64, 256, 8, 32), (1, 512, 128, 8, 32), (2, 24, 128, 8, 64)]:
        x = (torch.randn(n, cin, h, w, generator=gen) * 1.5).to(dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(cout, cin, 3, 3, generator=gen).to(dev)                        # raw N(0, 1) weights, as a float32 layer holds them
        st = (torch.rand(n, cin, generator=gen) + 0.5).to(dev)
        sc = (torch.rand(n, cout, generator=gen) * 0.05 + 0.01).to(dev)
        bias = (torch.randn(cout, generator=gen) * 0.2).to(dev)
        noise = (torch.randn(h, w, generator=gen) * 0.1).to(dev)
        x3 = gnerf_hip.split_f16x3(x, st)
        assert x3.shape == (n, 3 * cin, h, w) and x3.dtype == torch.float16 and gnerf_hip.is_channels_last(x3)
        xs = x * st[:, :, None, None]
        hi = x3[:, :cin].float()
        assert torch.equal(hi, xs.half().float()) and torch.equal(x3[:, 2 * cin:], x3[:, :cin]) and torch.equal(x3[:, cin:2 * cin].float(), (xs - hi).half().float())
        w3 = gnerf_hip.pack_conv3x3_weights_f32x3(wt)
        assert w3.shape == (9, cout, -(-3 * cin // 64) * 64)
        ref = torch.nn.functional.conv2d(xs.double(), wt.double(), padding=1)
        top = float(ref.abs().max())
        e_fw = float((torch.nn.functional.conv2d(xs, wt, padding=1).double() - ref).abs().max())
        got = gnerf_hip.conv3x3_f32x3_epilogue(x3, w3, alpha=1.0)
        assert got.shape == (n, cout, h, w) and got.dtype == torch.float32 and gnerf_hip.is_channels_last(got)
        e_own = float((got.double() - ref).abs().max())
        chain = 2.0 ** -24 * (9 * cin) ** 0.5                          # float32 chain over K = 9 cin products, relative to the largest partial sum
        assert e_own <= max(4 * e_fw, chain * top), ('plain', n, cin, cout, e_own, e_fw, top)
        # the whole layer: demodulation, noise, bias, lrelu * gain, clamp
        for scale, nz, clamp in ((sc, noise, None), (sc, None, 2.0), (None, noise, None)):
            got = gnerf_hip.conv3x3_f32x3_epilogue(x3, w3, bias, scale=scale, noise=nz, gain=1.3, clamp=clamp)
            t = ref * (scale[:, :, None, None].double() if scale is not None else 1.0) + (nz.double() if nz is not None else 0.0)
            want = bias_act.bias_act(t, bias.double(), act='lrelu', gain=1.3, clamp=clamp)
            e = float((got.double() - want).abs().max())
            amp = (float(scale.max()) if scale is not None else 1.0) * 1.3
            assert e <= max(4 * e_fw, chain * top) * amp + 1e-6 * float(want.abs().max()), ('layer', n, cin, cout, scale is not None, nz is not None, clamp, e, e_fw)
        # the stride-2 transposed form
        wp = gnerf_hip.pack_conv_transpose3x3_weights_f32x3(wt)
        ref_t = torch.nn.functional.conv_transpose2d(xs.double(), wt.transpose(0, 1).double(), stride=2)
        e_fw_t = float((torch.nn.functional.conv_transpose2d(xs, wt.transpose(0, 1).contiguous(), stride=2).double() - ref_t).abs().max())
        got_t = gnerf_hip.conv_transpose3x3_s2_f32x3(x3, wp)
        assert got_t.shape == (n, cout, 2 * h + 1, 2 * w + 1) and got_t.dtype == torch.float32
        e_t = float((got_t.double() - ref_t).abs().max())
        assert e_t <= max(4 * e_fw_t, 2.0 ** -24 * (4 * cin) ** 0.5 * float(ref_t.abs().max())), ('transposed', n, cin, cout, e_t, e_fw_t)
    assert int(gnerf_hip.split_overflow_flag(dev).item()) == 0
    # an operand outside float16's range: saturated, finite, and REPORTED
    big = torch.full((1, 8, 8, 32), 1e5, device=dev).contiguous(memory_format=torch.channels_last)
    b3 = gnerf_hip.split_f16x3(big)
    assert torch.isfinite(b3.float()).all() and int(gnerf_hip.split_overflow_flag(dev).item()) == 1
    gnerf_hip.split_overflow_flag(dev).zero_()


def test_generator_fast_modconv_path_equals_plain_path(dev, monkeypatch):
    """The generator with csrc/modconv.hip around its convolutions (and the shared-weight convolution form for fp16 batches)
    against the same generator on the plain PyTorch-op chains: batch 1 (grouped form) and batch 3 (shared-weight form)."""
    import gnerf_generator as GG
    import gnerf_harness as H
    torch.manual_seed(1)
    G = GG.Generator().eval().requires_grad_(False).to(dev)
    with torch.no_grad():
        for n_, p_ in G.named_parameters():
            if n_.endswith('noise_strength') or n_.endswith('.bias'):
                p_.add_(torch.randn_like(p_) * 0.1)
            if n_.endswith('torgb.weight') and 'superresolution' in n_:
                p_.mul_(0.08)
        for nb in (1, 3):
            z = torch.randn(nb, 512, device=dev)
            c = torch.cat([H.camera_label(H.orbit_pose(5 + 9 * i, 120)) for i in range(nb)]).to(dev)
            ws = G.mapping(z, c)
            outs = []
            for fast in (True, False):
                monkeypatch.setattr(GG, '_MODCONV_FAST', fast)
                torch.manual_seed(3)
                outs.append(G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=64))
            for k in ('image', 'image_raw', 'image_depth'):
                mse = float(((outs[0][k] - outs[1][k]) ** 2).mean())
                assert mse < (1e-5 if k != 'image_depth' else 1e-6), (nb, k, mse)


def test_frozen_generator_passes_gradient_to_latent(dev, monkeypatch):
    """The reference's encoder phase (training_loop.py:172,318-322): G frozen (`requires_grad_(False)`), the latent `ws` produced by a
    trainable identity encoder.  The const block's input carries no gradient then, so the fast-path predicate must look at the
    LATENT: with csrc/modconv.hip's kernels (which detach the styles) the gradient image -> ws would be cut silently.  dImage/dws of
    the default flow must equal the plain PyTorch-op flow's, and be non-zero for every layer's slice of ws."""
    import gnerf_generator as GG
    import gnerf_harness as H
    torch.manual_seed(4)
    G = GG.Generator().eval().requires_grad_(False).to(dev)
    z = torch.randn(2, 512, device=dev)
    c = torch.cat([H.camera_label(H.orbit_pose(3 + 13 * i, 120)) for i in range(2)]).to(dev)
    with torch.no_grad():
        ws0 = G.mapping(z, c)
    probe = torch.randn(2, 3, 512, 512, device=dev)
    grads, images = [], []
    for fast in (True, False):
        monkeypatch.setattr(GG, '_MODCONV_FAST', fast)
        ws = ws0.clone().requires_grad_(True)
        torch.manual_seed(6)
        out = G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=32)
        assert out['image'].requires_grad and out['image_raw'].requires_grad, fast
        loss = (out['image'].float() * probe).mean() + out['image_raw'].float().mean() + out['image_depth'].mean()
        (g,) = torch.autograd.grad(loss, ws)
        grads.append(g)
        images.append(out['image'].detach().float())
    assert torch.isfinite(grads[0]).all()
    per_layer = grads[0].abs().amax(dim=(0, 2))
    assert (per_layer > 0).all(), per_layer                                        # every ws slice reaches the image through some layer
    rel = float((grads[0] - grads[1]).norm() / grads[1].norm())
    assert rel < 1e-4, rel                                                         # both settings take the same autograd forms here (float atomics reorder)
    assert float(((images[0] - images[1]) ** 2).mean()) < 1e-10
    # and without a gradient on the latent the fast kernels are still what runs (the predicate did not get stricter than needed)
    monkeypatch.setattr(GG, '_MODCONV_FAST', True)
    calls = []
    import gnerf_hip
    orig = gnerf_hip.modconv_epilogue
    monkeypatch.setattr(gnerf_hip, 'modconv_epilogue', lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    with torch.no_grad():
        G.synthesis(ws0, c, noise_mode='const', neural_rendering_resolution=32)
    assert calls


def test_generator_synthesis_under_inference_mode(dev):
    """G.synthesis under torch.inference_mode(): the channels_last plane producer's output is an inference tensor there (no
    version counter -> it goes untagged and the render launcher measures max |planes| itself), and the result must be the
    torch.no_grad() one bit for bit (same kernels, same uniform draws)."""
    import gnerf_generator as GG
    import gnerf_harness as H
    torch.manual_seed(2)
    G = GG.Generator().eval().requires_grad_(False).to(dev)
    z = torch.randn(2, 512, device=dev)
    c = torch.cat([H.camera_label(H.orbit_pose(7 + 11 * i, 120)) for i in range(2)]).to(dev)
    outs = []
    for ctx in (torch.no_grad, torch.inference_mode):
        with ctx():
            torch.manual_seed(5)
            outs.append(G.synthesis(G.mapping(z, c), c, noise_mode='const', neural_rendering_resolution=64))
    assert outs[1]['image'].is_inference() and not outs[0]['image'].is_inference()
    for k in ('image', 'image_raw', 'image_depth'):
        assert torch.equal(outs[0][k], outs[1][k].clone()), k


# ---- the whole generator around the hot path (callers in PyTorch/MIOpen, renderer + ops native) ----------------------

def test_generator_forward_gpu_vs_cpu(dev):
    """BASELINE config 3 at batch 1: gnerf_generator.Generator on the GPU (fused renderer, native bias_act / upfirdn2d, fp16
    superresolution) against the same module on the CPU (PyTorch-op renderer and ops, fp32) with the same weights, latent,
    camera and the same two uniform draws.  Tolerance = north_star's: pixel MSE < 1e-4 on images in [-1, 1]."""
    import copy
    import gnerf_generator
    import gnerf_harness as H
    from test_host_cpu import _Replay
    torch.manual_seed(3)
    G = gnerf_generator.Generator().eval().requires_grad_(False)
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for n, p in G.named_parameters():
            if n.endswith('noise_strength') or n.endswith('.bias'):
                p.add_(torch.randn(p.shape, generator=gen) * 0.1)
            if n.endswith('torgb.weight') and 'superresolution' in n:
                p.mul_(0.08)                                  # random-init images inside about [-1, 1]: the tolerance below is absolute
    z = torch.randn(1, 512, generator=gen)
    c = H.camera_label(H.orbit_pose(17, 120))
    res, S = 64, 48
    draws = [torch.rand(1, res * res, S, 1, generator=gen), torch.rand(res * res, S, generator=gen)]
    with torch.no_grad():
        ws = G.mapping(z, c)
        with _Replay([d.clone() for d in draws]):
            ref = G.synthesis(ws, c, neural_rendering_resolution=res, noise_mode='const')
        Gd = copy.deepcopy(G).to(dev)
        with _Replay([d.to(dev) for d in draws]):
            out = Gd.synthesis(Gd.mapping(z.to(dev), c.to(dev)), c.to(dev), neural_rendering_resolution=res, noise_mode='const')
    assert out['image'].shape == (1, 3, 512, 512) and out['image'].dtype == torch.float32
    scale = float(ref['image'].abs().max())
    assert 0.5 < scale < 3.0, scale
    for k in ('image', 'image_raw'):
        mse = float(((out[k].cpu() - ref[k]) ** 2).mean())
        assert mse < 1e-4, (k, mse, scale)
    assert float((out['image_depth'].cpu() - ref['image_depth']).abs().max()) < 2e-3


def test_config3_n4_gpu_vs_reference_fixture(dev, golden, monkeypatch):
    """BASELINE config 3 as SURVEY section 8d defines it: the full generator forward at N=4, render resolution 64, constant
    noise, on the GPU (fused renderer, native ops, fp16 superresolution) against the fixture made from the REFERENCE's
    TriPlaneGenerator on the CPU in fp32 (tests/golden/make_golden.py; same weights, noise and batch through det_init).
    Tolerance: north_star's pixel MSE < 1e-4, un-normalised (the fixture's images have std 0.35, |max| 1.6); the fp32 run of
    the same GPU path is held to 1e-7."""
    import gen_cases as C
    g = golden('generator_n4.npz')
    G, _ = C.build(dev)
    import gnerf_hip
    import gnerf_generator as GG
    own = {'conv3x3_epilogue': [], 'conv_transpose3x3_s2': [], 'conv3x3_epilogue_torgb': []}
    real_fns = {name: getattr(gnerf_hip, name) for name in own}
    for name in own:
        monkeypatch.setattr(gnerf_hip, name, (lambda real_fn, name: lambda x, w, *a, **kw: (own[name].append((x.shape[0], x.shape[1], w.shape[1], x.shape[2])), real_fn(x, w, *a, **kw))[1])(real_fns[name], name))
    own32 = {'conv3x3_f32x3_epilogue': [], 'conv_transpose3x3_s2_f32x3': []}
    for name in own32:
        monkeypatch.setattr(gnerf_hip, name, (lambda real_fn, name: lambda x, w, *a, **kw: (own32[name].append((x.shape[0], x.shape[1], w.shape[1], x.shape[2])), real_fn(x, w, *a, **kw))[1])(getattr(gnerf_hip, name), name))
    gnerf_hip.split_overflow_flag(dev).zero_()
    for force_fp32, tol in ((False, 1e-4), (True, 1e-7)):
        for name in own:
            own[name].clear()
        for name in own32:
            own32[name].clear()
        ws, out = C.run_config3(G, dev, **(dict(force_fp32=True) if force_fp32 else {}))
        # round 6: the ROUTE is part of the test.  fp16 superresolution: every 3x3 layer the kernel's shape gate admits runs on csrc/conv3x3.hip
        # (block0.conv1, block1.conv1 fused with their epilogues; block0.conv0, block1.conv0 as the transposed form); fp32: none does.
        if force_fp32:
            assert not own['conv3x3_epilogue'] and not own['conv_transpose3x3_s2'] and not own['conv3x3_epilogue_torgb'], own
            # ... the float32 layers from 64^2 up run in fp32-grade arithmetic on the same kernel (gnerf_hip.conv3x3_f32x3_epilogue: x3 has three
            # times the layer's input channels): backbone b64.conv1, b128.conv1, b256.conv1 and the superresolution's block0.conv1, block1.conv1;
            # the x2 layers b64.conv0 ... b256.conv0, block0.conv0, block1.conv0 on its transposed form
            assert sorted(own32['conv3x3_f32x3_epilogue']) == [(4, 384, 128, 256), (4, 384, 128, 512), (4, 768, 256, 128), (4, 768, 256, 256), (4, 1536, 512, 64)], own32
            assert sorted(own32['conv_transpose3x3_s2_f32x3']) == [(4, 96, 256, 128), (4, 768, 128, 128), (4, 768, 128, 256), (4, 1536, 256, 64), (4, 1536, 512, 32)], own32
            assert int(gnerf_hip.split_overflow_flag(dev).item()) == 0
        else:
            # (the LAST block's conv1 with the block's ToRGB in its epilogue, no layer output: gnerf_conv3x3_epilogue_torgb_nhwc)
            assert sorted(own['conv3x3_epilogue']) == [(4, 256, 256, 256)] and own['conv3x3_epilogue_torgb'] == [(4, 128, 128, 512)], own
            assert sorted(own['conv_transpose3x3_s2']) == [(4, 32, 256, 128), (4, 256, 128, 256)], own
            # (the backbone is float32 in both legs: its six layers from 64^2 up)
            assert sorted(own32['conv3x3_f32x3_epilogue']) == [(4, 384, 128, 256), (4, 768, 256, 128), (4, 1536, 512, 64)], own32
            assert sorted(own32['conv_transpose3x3_s2_f32x3']) == [(4, 768, 128, 128), (4, 1536, 256, 64), (4, 1536, 512, 32)], own32
        np.testing.assert_allclose(ws[:, 0, :8].cpu().numpy(), g['ws_first'], atol=1e-4)
        assert out['image'].shape == (4, 3, 512, 512) and out['image'].dtype == torch.float32
        mse = {'image': float(((out['image'][:, :, 4::8, 4::8].cpu().numpy() - g['image_sub']) ** 2).mean()),
               'image_raw': float(((out['image_raw'].cpu().numpy() - g['image_raw']) ** 2).mean())}
        assert mse['image'] < tol and mse['image_raw'] < tol, (force_fp32, mse)
        np.testing.assert_allclose(out['image_depth'].cpu().numpy(), g['image_depth'], atol=1e-3)
        np.testing.assert_allclose(out['image'].mean((1, 2, 3)).cpu().numpy(), g['image_mean'], atol=2e-3)
    b = C.batch_on(dev)
    with torch.no_grad(), C.DI.DetNoise('config3'):
        d = G.synthesis(G.mapping(b['z'], b['c']), b['c'], noise_mode='const', neural_rendering_resolution=64, only_depth=True)
    assert d['image'] is d['image_depth'] and torch.equal(d['image'], out['image_depth'])
    # GNERF_FUSED_CONV=0 (MIOpen + the stand-alone epilogue) gives the same fp16 image within the fixture's tolerance, and takes the other route
    monkeypatch.setattr(GG, '_FUSED_CONV', False)
    for name in own:
        own[name].clear()
    _, out_plain = C.run_config3(G, dev)
    assert not own['conv3x3_epilogue'] and not own['conv_transpose3x3_s2'] and not own['conv3x3_epilogue_torgb'], own
    assert float(((out_plain['image'][:, :, 4::8, 4::8].cpu().numpy() - g['image_sub']) ** 2).mean()) < 1e-4
    _, out_fused = (monkeypatch.setattr(GG, '_FUSED_CONV', True), C.run_config3(G, dev))[1]
    assert float(((out_plain['image'] - out_fused['image']) ** 2).mean()) < 1e-5
    # GNERF_FUSED_TORGB=0: the last block as three launches (layer, ToRGB into the image) -- the same roundings, so the same image up to the order
    # of an fp32 sum in front of an fp16 rounding
    monkeypatch.setattr(GG, '_FUSED_TORGB', False)
    for name in own:
        own[name].clear()
    _, out_three = C.run_config3(G, dev)
    assert sorted(own['conv3x3_epilogue']) == [(4, 128, 128, 512), (4, 256, 256, 256)] and not own['conv3x3_epilogue_torgb'], own
    assert float((out_three['image'] - out_fused['image']).abs().max()) < 4e-3 and float(((out_three['image'] - out_fused['image']) ** 2).mean()) < 1e-8
    monkeypatch.setattr(GG, '_FUSED_TORGB', True)
    # a batch of ONE takes the shared-weight form and the same kernels (the frame-by-frame orbit)
    for name in own:
        own[name].clear()
    with torch.no_grad(), C.DI.DetNoise('config3'):
        one = G.synthesis(G.mapping(b['z'][:1], b['c'][:1]), b['c'][:1], noise_mode='const', neural_rendering_resolution=64)
    assert sorted(own['conv3x3_epilogue']) == [(1, 256, 256, 256)] and own['conv3x3_epilogue_torgb'] == [(1, 128, 128, 512)] and len(own['conv_transpose3x3_s2']) == 2, own
    assert float(((one['image'][:, :, 4::8, 4::8].cpu().numpy() - g['image_sub'][:1]) ** 2).mean()) < 1e-4


def test_config3_reference_flow_through_overlay_gpu_vs_fixture(dev, golden, monkeypatch):
    """The flow a G-NeRF checkout runs after the swap: the layer code of the reference (gnerf_generator's plain path, GNERF_MODCONV_FAST=0:
    PyTorch-op weight modulation, `modulated_conv2d`'s calls into torch_utils.ops.conv2d_resample and bias_act / upfirdn2d) with the
    OVERLAY's modules on the GPU -- fp16 superresolution blocks convolved channels_last, ToRGB on the streaming kernel, NCHW planes into
    the fused renderer -- against the fixture made from the reference's TriPlaneGenerator on the CPU (config 3, N=4 and N=1)."""
    import gen_cases as C
    import gnerf_generator as GG
    from torch_utils.ops import conv2d_resample as CR
    g = golden('generator_n4.npz')
    G, _ = C.build(dev)
    monkeypatch.setattr(GG, '_MODCONV_FAST', False)
    G.backbone.synthesis.b256.emit_channels_last = False
    seen = {'calls': 0, 'channels_last_out': 0, 'torgb_kernel': 0}
    real = CR.conv2d_resample

    def spy(x, w, **kw):
        y = real(x=x, w=w, **kw)
        seen['calls'] += 1
        seen['channels_last_out'] += int(y.dtype == torch.float16 and y.shape[1] > 3 and CR._is_channels_last(y))
        seen['torgb_kernel'] += int(y.dtype == torch.float16 and y.shape[1] == 3 and x.shape[0] == 1 and CR._is_channels_last(x))
        return y
    monkeypatch.setattr(CR, 'conv2d_resample', lambda x, w, **kw: spy(x, w, **kw))
    ws, out = C.run_config3(G, dev)
    assert seen['calls'] == 29 and seen['channels_last_out'] == 0, seen         # every modulated convolution and ToRGB; a batch keeps the reference's layout
    mse = {'image': float(((out['image'][:, :, 4::8, 4::8].cpu().numpy() - g['image_sub']) ** 2).mean()),
           'image_raw': float(((out['image_raw'].cpu().numpy() - g['image_raw']) ** 2).mean())}
    assert mse['image'] < 1e-4 and mse['image_raw'] < 1e-4, mse
    np.testing.assert_allclose(out['image_depth'].cpu().numpy(), g['image_depth'], atol=1e-3)
    # batch 1 (an orbit frame): the grouped convolution degenerates to a plain one, x stays channels_last into ToRGB's streaming kernel
    b = C.batch_on(dev)
    seen.update(calls=0, channels_last_out=0, torgb_kernel=0)
    with torch.no_grad(), C.DI.DetNoise('config3'):
        one = G.synthesis(G.mapping(b['z'][:1], b['c'][:1]), b['c'][:1], noise_mode='const', neural_rendering_resolution=64)
    assert seen['calls'] == 29 and seen['channels_last_out'] == 6 and seen['torgb_kernel'] == 3, seen      # the six fp16 convolutions stay channels_last
    assert float(((one['image'][:, :, 4::8, 4::8].cpu().numpy() - g['image_sub'][:1]) ** 2).mean()) < 1e-4
    np.testing.assert_allclose(one['image_depth'].cpu().numpy(), g['image_depth'][:1], atol=1e-3)
    # round 6: of those six, the four 3x3 layers of the 256^2 / 512^2 blocks (32 -> 256 x2, 256 -> 256, 256 -> 128 x2, 128 -> 128) run on
    # csrc/conv3x3.hip -- the call gen_videos.py makes, one camera per synthesis (gen_videos.py:154-171); block64's two 32 -> 32 layers
    # are outside the kernel's 128-output-channel blocks and stay with the framework.  Spied at the library's Python entry points.
    import gnerf_hip
    own = {'conv3x3_epilogue': [], 'conv_transpose3x3_s2': []}
    for name in own:
        real_fn = getattr(gnerf_hip, name)
        monkeypatch.setattr(gnerf_hip, name, (lambda real_fn, name: lambda x, w, *a, **kw: (own[name].append((x.shape[1], w.shape[1], x.shape[2])), real_fn(x, w, *a, **kw))[1])(real_fn, name))
    with torch.no_grad(), C.DI.DetNoise('config3'):
        again = G.synthesis(G.mapping(b['z'][:1], b['c'][:1]), b['c'][:1], noise_mode='const', neural_rendering_resolution=64)
    assert sorted(own['conv3x3_epilogue']) == [(128, 128, 512), (256, 256, 256)] and sorted(own['conv_transpose3x3_s2']) == [(32, 256, 128), (256, 128, 256)], own
    assert float(((again['image'] - one['image']) ** 2).mean()) < 1e-5          # (the renderer's uniform draws differ from call to call)
    # ... and with the route switched off (GNERF_FUSED_CONV=0: MIOpen convolves) the frame is the same image within the fixture's tolerance
    monkeypatch.setattr(CR, '_FUSED_CONV', False)
    for name in own:
        own[name].clear()
    with torch.no_grad(), C.DI.DetNoise('config3'):
        plain = G.synthesis(G.mapping(b['z'][:1], b['c'][:1]), b['c'][:1], noise_mode='const', neural_rendering_resolution=64)
    assert not own['conv3x3_epilogue'] and not own['conv_transpose3x3_s2']
    assert float(((plain['image'] - one['image']) ** 2).mean()) < 1e-5
    assert float(((plain['image'][:, :, 4::8, 4::8].cpu().numpy() - g['image_sub'][:1]) ** 2).mean()) < 1e-4


def test_config5_training_step_gpu_vs_reference_fixture(dev, golden):
    """BASELINE config 5's step on the GPU -- G in training mode (un-fused modulated convolutions, random backbone noise) through
    the fused renderer's forward AND backward kernels, L1 + L1 + 1.2 softplus(-D(depth)), then the D step with R1 (a double
    backward through bias_act / upfirdn2d's second-order forms) -- against the fixture made from the REFERENCE's classes on the
    CPU: every loss term and the gradient norm of every parameter of G and D.  fp32 everywhere first (tight), then with the fp16
    superresolution / discriminator blocks the reference uses on a GPU (loose)."""
    import gen_cases as C
    import train_step_mi355x as T
    g = golden('train_step.npz')
    G, D = C.build(dev)
    parts, gen, g_norms, d_norms = C.run_config5(G, D, dev, force_fp32=True)
    assert abs(parts['loss'] - float(g['loss'])) < 2e-4 * abs(float(g['loss'])), parts
    assert abs(parts['gan'] - float(g['loss_gan'])) < 2e-4 and abs(parts['d_gen'] - float(g['loss_dgen'])) < 2e-4
    assert abs(parts['d_real'] - float(g['loss_dreal'])) < 2e-4
    assert abs(parts['d_r1'] - float(g['loss_r1'].mean())) < 5e-3 * float(g['loss_r1'].mean())
    np.testing.assert_allclose(gen['image_raw'].detach().cpu().numpy(), g['image_raw'], atol=1e-3)
    np.testing.assert_allclose(gen['image_depth'].detach().cpu().numpy(), g['image_depth'], atol=1e-3)
    C.compare_norms(g_norms, g['g_names'], g['g_grad_norms'], 2e-2, 'G')
    C.compare_norms(d_norms, g['d_names'], g['d_grad_norms'], 2e-2, 'D')
    w1g = G.decoder.net[0].weight.grad.cpu()
    assert _rel_l2(w1g, torch.from_numpy(g['g_grad_decoder_w1'])) < 5e-3
    # three whole tensors elementwise (backbone, superresolution, discriminator): relative L2 <= 1e-3 in fp32
    C.compare_whole_gradients(G, D, golden('train_step_grads.npz'), 1e-3, 'gpu fp32')
    # the reference's GPU precision policy: fp16 superresolution and discriminator blocks
    parts16, gen16, g16, d16 = C.run_config5(G, D, dev, force_fp32=False)
    assert gen16['image'].dtype == torch.float32 and all(np.isfinite(v) for v in parts16.values())
    assert abs(parts16['loss'] - float(g['loss'])) < 2e-2 * abs(float(g['loss'])), parts16
    tot = lambda d: float(np.sqrt(sum(v * v for v in d.values())))                                        # noqa: E731
    assert abs(tot(g16) - float(np.sqrt((g['g_grad_norms'] ** 2).sum()))) < 0.1 * tot(g16)
    assert abs(tot(d16) - float(np.sqrt((g['d_grad_norms'] ** 2).sum()))) < 0.1 * tot(d16)
    # and the whole step (exchange + both optimisers) runs and moves both networks
    opt_G = torch.optim.Adam(G.parameters(), lr=1e-3, betas=(0.9, 0.999))
    opt_D = torch.optim.Adam(D.parameters(), lr=1e-3, betas=(0.0, 0.99))
    before = (G.decoder.net[0].weight.detach().clone(), D.b4.out.weight.detach().clone())
    G.train()
    for _ in range(2):
        out = T.gd_train_step(G, D, opt_G, opt_D, C.batch_on(dev))
    assert all(np.isfinite(float(v)) for v in out.values())
    assert not torch.equal(before[0], G.decoder.net[0].weight) and not torch.equal(before[1], D.b4.out.weight)
    assert not any(p.requires_grad for p in G.parameters()) and not any(p.requires_grad for p in D.parameters())


@pytest.mark.production_path
def test_render_backward_full_size_properties(dev):
    """gnerf_render_backward at the training shape of BASELINE config 5 (4 items x 64x64 rays, 48+48 samples, 256x256 planes),
    too big for the float64 oracle in a unit test: finite; deterministic up to the order of its float atomics (repeat runs
    agree to 1e-5 of the largest entry, decoder gradients to 1e-5 relative); a planes-only request gives the same plane gradient
    as the full request and a decoder-only request the same decoder gradients; the gradient is linear in the incoming gradient;
    rays that receive zero gradient contribute nothing (item independence); and a strided subset of rays against autograd
    through the float64 oracle."""
    import gnerf_hip
    N, res, S, F = 4, 64, 48, 48
    planes, dec, o, d, nc, nf = _random_scene(3, N=N, res=res, S=S, F=F, hw=(256, 256), scale=1.0)
    M = res * res
    gen = torch.Generator().manual_seed(9)
    g_rgb, g_depth, g_wsum = torch.randn(N, M, 32, generator=gen), torch.randn(N, M, 1, generator=gen), torch.randn(N, M, 1, generator=gen)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    de = [t.to(dev) for t in dec]
    args = (nhwc, N, de, o.to(dev), d.to(dev), nc.to(dev), nf.to(dev))
    kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)
    gr, gd_, gw = g_rgb.to(dev), g_depth.to(dev), g_wsum.to(dev)
    gp, gdec = gnerf_hip.render_backward(*args, gr, gd_, gw, **kw)
    assert torch.isfinite(gp).all() and all(torch.isfinite(t).all() for t in gdec)
    assert float(gp.abs().max()) > 0
    gp2, gdec2 = gnerf_hip.render_backward(*args, gr, gd_, gw, **kw)
    assert float((gp - gp2).abs().max()) <= 1e-5 * float(gp.abs().max())
    for a, b in zip(gdec, gdec2):
        assert _rel(a, b) < 1e-5
    gp_only, none = gnerf_hip.render_backward(*args, gr, gd_, gw, need_decoder=False, **kw)
    assert none is None and float((gp_only - gp).abs().max()) <= 1e-5 * float(gp.abs().max())
    none, dec_only = gnerf_hip.render_backward(*args, gr, gd_, gw, need_planes=False, **kw)
    assert none is None and all(_rel(a, b) < 1e-5 for a, b in zip(dec_only, gdec))
    # linear in the incoming gradient
    gp3, gdec3 = gnerf_hip.render_backward(*args, gr * 3, gd_ * 3, gw * 3, **kw)
    assert float((gp3 - 3 * gp).abs().max()) <= 2e-5 * float(gp3.abs().max())
    # item independence: only item 2 receives a gradient -> only its three planes get one, equal to a batch of that item alone
    mask = torch.zeros(N, 1, 1, device=dev)
    mask[2] = 1
    gp_m, _ = gnerf_hip.render_backward(*args, gr * mask, gd_ * mask, gw * mask, need_decoder=False, **kw)
    assert float(gp_m[:6].abs().max()) == 0 and float(gp_m[9:].abs().max()) == 0
    one, _ = gnerf_hip.render_backward(nhwc[6:9].contiguous(), 1, de, args[3][2:3], args[4][2:3], args[5][2:3], args[6][2 * M:3 * M],
                                       gr[2:3], gd_[2:3], gw[2:3], need_decoder=False, **kw)
    assert float((one - gp_m[6:9]).abs().max()) <= 1e-5 * float(one.abs().max())
    # a strided subset of rays against the float64 oracle: zero the incoming gradient everywhere else
    idx = torch.arange(0, M, 997)
    sel = torch.zeros(N, M, 1)
    sel[:, idx] = 1
    gp_s, gdec_s = gnerf_hip.render_backward(*args, gr * sel.to(dev), gd_ * sel.to(dev), gw * sel.to(dev), **kw)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
    sub_nf = nf.reshape(N, M, F)[:, idx].reshape(-1, F)
    ref_planes, ref_dec = _oracle_grads(planes, dec, o[:, idx], d[:, idx], nc[:, idx], sub_nf, opts, g_rgb[:, idx], g_depth[:, idx], g_wsum[:, idx])
    gp_nchw = gp_s.reshape(N, 3, 256, 256, 32).permute(0, 1, 4, 2, 3).cpu()
    assert _rel(gp_nchw, ref_planes) < 2e-3 and _rel_l2(gp_nchw, ref_planes) < 1e-3, (_rel(gp_nchw, ref_planes), _rel_l2(gp_nchw, ref_planes))
    for name, a, b in zip(['w1', 'b1', 'w2', 'b2'], gdec_s, ref_dec):
        assert _rel_l2(a.cpu(), b) < 1e-3, (name, _rel_l2(a.cpu(), b))


def test_generator_per_latent_constants_are_cached_and_invalidated(dev):
    """gen_videos.py renders an orbit from ONE ws (gen_videos.py:150): style vectors, modulated weights and demodulation coefficients are
    constants of it and are cached per layer (gnerf_generator._per_latent).  A second frame must reuse them (no modulate_weights launch),
    and a changed latent -- another tensor, or the same one modified in place -- or a changed parameter must not."""
    import gnerf_generator as GG
    import gnerf_harness as H
    import gnerf_hip
    torch.manual_seed(6)
    G = GG.Generator().eval().requires_grad_(False).to(dev)
    c = H.camera_label(H.orbit_pose(3, 120)).to(dev)
    calls = {'n': 0}
    real = gnerf_hip.modulate_weights

    def counting(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    gnerf_hip.modulate_weights = counting
    try:
        with torch.no_grad():
            ws = G.mapping(torch.randn(1, 512, device=dev), c)
            run = lambda w: G.synthesis(w, c, noise_mode='const', neural_rendering_resolution=64)
            # (MIOpen's convolutions are not bit-reproducible from call to call here -- 7e-3 on these images -- hence the tolerances)
            close = lambda p, q: float((p['image'] - q['image']).abs().max()) < 3e-2
            torch.manual_seed(1); a = run(ws); first = calls['n']
            torch.manual_seed(1); b = run(ws); second = calls['n'] - first
            assert first > 0 and second == 0 and close(a, b)
            ws2 = ws + 0.5 * torch.randn_like(ws[:, :1])             # another latent: everything is recomputed, and the image differs
            torch.manual_seed(1); d = run(ws2)
            assert calls['n'] - first == first and not close(d, a)
            base = calls['n']
            ws.copy_(ws2)                                            # the SAME tensor modified in place: its version counter invalidates
            torch.manual_seed(1); e = run(ws)
            assert calls['n'] - base == first and close(e, d)
            base = calls['n']
            w1 = G.superresolution.block1.conv1.weight
            w1.add_(torch.randn_like(w1))                            # a parameter modified in place: that layer alone is recomputed
            torch.manual_seed(1); f = run(ws)
            assert calls['n'] - base == 1 and not close(f, e)
    finally:
        gnerf_hip.modulate_weights = real


def test_density_volume_gpu_vs_cpu(dev):
    """gen_videos.py --shapes counterpart: the density lattice through the fused point-query kernel (chunked, ragged last chunk)
    against the same module's PyTorch-op path on the CPU."""
    import copy
    import gnerf_generator
    import gen_videos_mi355x as gv
    torch.manual_seed(5)
    G = gnerf_generator.Generator().eval().requires_grad_(False)
    with torch.no_grad():
        ws = G.mapping(torch.randn(1, 512), torch.zeros(1, 25))
        ref = gv.extract_density_grid(G, ws, resolution=40, max_batch=64000, crop=True)
        Gd = copy.deepcopy(G).to(dev)
        vol = gv.extract_density_grid(Gd, ws.to(dev), resolution=40, max_batch=7001, crop=True)
    assert vol.shape == (40, 40, 40) and vol.is_cuda
    scale = float(ref.abs().max())
    assert scale > 0.1
    assert float((vol.cpu() - ref).abs().max()) < 2e-4 * max(1.0, scale)


def test_frame_program_replay_equals_eager(dev):
    """gen_videos harness: an orbit frame replayed from the captured HIP graph (FrameProgram) is the frame plain launches give
    for the same camera and the same generator state, bit for bit (uint8 images and the raw 64x64 image), replay after replay."""
    import gnerf_generator
    import gnerf_harness as H
    import gen_videos_mi355x as gv
    torch.manual_seed(2)
    with torch.no_grad():
        G = gnerf_generator.Generator().eval().requires_grad_(False).to(dev)
        G.rendering_kwargs['depth_resolution'] = G.rendering_kwargs['depth_resolution_importance'] = 96      # the CLI's doubled sampling
        z = torch.randn(1, 512, device=dev)
        ws = gv.orbit_latents(G, z, dev)
        prog = gv.FrameProgram(G, ws, 64, dev)
        for i in (3, 77):
            c = H.camera_label(H.orbit_pose(i, 240, device=dev))
            torch.manual_seed(100 + i)
            f_graph, r_graph = prog(c)
            torch.manual_seed(100 + i)
            out = G.synthesis(ws=ws, c=c, noise_mode='const', neural_rendering_resolution=64, use_cached_backbone=True)
            assert torch.equal(f_graph, H.to_uint8(out['image'])) and torch.equal(r_graph, H.to_uint8(out['image_raw']))
        assert f_graph.shape == (1, 512, 512, 3) and f_graph.dtype == torch.uint8 and float(f_graph.float().std()) > 1.0
        # The program owns what its graph reads: an eager frame of ANOTHER latent with cache_backbone=True replaces the
        # generator's cached planes and the renderer's NHWC copy (and frees the old ones); the next replay must still render
        # the program's own latent, not recycled memory.
        c = H.camera_label(H.orbit_pose(5, 240, device=dev))
        torch.manual_seed(7)
        want, want_raw = prog(c)
        ws2 = gv.orbit_latents(G, torch.randn(1, 512, device=dev), dev)
        other = G.synthesis(ws=ws2, c=c, noise_mode='const', neural_rendering_resolution=64, cache_backbone=True)
        junk = [torch.randn_like(G._last_planes) for _ in range(4)]          # recycle whatever the allocator got back
        assert not torch.equal(H.to_uint8(other['image']), want)
        torch.manual_seed(7)
        again, again_raw = prog(c)
        assert torch.equal(again, want) and torch.equal(again_raw, want_raw)
        torch.manual_seed(7)                                                  # ... and eager frames after a replay see the program's planes
        out = G.synthesis(ws=ws, c=c, noise_mode='const', neural_rendering_resolution=64, use_cached_backbone=True)
        assert torch.equal(H.to_uint8(out['image']), want)
        del junk


# ---- grid_sample_gradfix on the native sampler -----------------------------------------------------------------------

@pytest.mark.parametrize('dtype', ['float32', 'float16'])
@pytest.mark.parametrize('shape', [((2, 3, 7, 9), (5, 6)), ((1, 70, 33, 20), (64, 64)), ((3, 1, 4, 4), (1, 300))])
def test_grid_sample_native_vs_oracle(dev, dtype, shape):
    import gnerf_hip
    from oracle import ops_ref as O
    (N, C, H, W), (Ho, Wo) = shape
    g = torch.Generator().manual_seed(1)
    img = torch.randn(N, C, H, W, generator=g)
    grid = torch.rand(N, Ho, Wo, 2, generator=g) * 2.8 - 1.4           # a good share of the points fall outside the image
    grid[0, 0, 0] = torch.tensor([float('nan'), 0.0])
    grid[0, 0, -1] = torch.tensor([1e30, -1e30])
    td = getattr(torch, dtype)
    x = img.to(dev).to(td)
    assert gnerf_hip.grid_sample_supported(x, grid.to(dev))
    out = gnerf_hip.grid_sample_2d(x, grid.to(dev))
    ref = O.grid_sample_2d(x.float().cpu().numpy(), torch.nan_to_num(grid, nan=-9.0).clamp(-9, 9).numpy())      # NaN / huge = far outside = zeros
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == 'float32' else dict(rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, **tol)
    # channels_last image: same values through the strided path
    out_cl = gnerf_hip.grid_sample_2d(x.contiguous(memory_format=torch.channels_last), grid.to(dev))
    assert torch.equal(out_cl, out)


def test_grid_sample_gradfix_gpu_all_orders(dev):
    """The drop-in op with `enabled`: forward, first-order gradients (image and grid) and the second-order gradient w.r.t. the
    incoming gradient, on the native kernels, against autograd through torch's sampler on the CPU (the reference's path)."""
    from torch_utils.ops import grid_sample_gradfix
    g = torch.Generator().manual_seed(2)
    img, grid = torch.randn(2, 5, 11, 8, generator=g), torch.rand(2, 9, 7, 2, generator=g) * 2.4 - 1.2
    w1, w2 = torch.randn(2, 5, 9, 7, generator=g), torch.randn(2, 5, 11, 8, generator=g)

    def run(image, grd, sample, second_order):
        image, grd = image.clone().requires_grad_(True), grd.clone().requires_grad_(True)
        out = sample(image, grd)
        probe = w1.to(out.device).clone().requires_grad_(True)
        g_img, g_grid = torch.autograd.grad(out, (image, grd), probe)
        if second_order:                # (second order w.r.t. the grid is refused, upstream too: the grid is a constant in this pass)
            image2 = image.detach().clone().requires_grad_(True)
            (g_img2,) = torch.autograd.grad(sample(image2, grd.detach()), image2, probe, create_graph=True)
            (gg,) = torch.autograd.grad((g_img2 * w2.to(out.device)).sum(), probe)     # d/d(grad_out) of the image gradient
        else:       # torch's own sampler has no double backward (the reason the op exists); the image gradient is linear in the
            gg = sample(w2, grd)        # incoming gradient with the sampler as its transpose, so the derivative is sample(w2, grid)
        return [t.detach().cpu() for t in (out, g_img, g_grid, gg)]

    ref = run(img, grid, lambda a, b: torch.nn.functional.grid_sample(a, b, mode='bilinear', padding_mode='zeros', align_corners=False), False)
    grid_sample_gradfix.enabled = True
    try:
        got = run(img.to(dev), grid.to(dev), grid_sample_gradfix.grid_sample, True)
    finally:
        grid_sample_gradfix.enabled = False
    for a, b, name in zip(got, ref, ('out', 'grad_image', 'grad_grid', 'grad_grad_out')):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-4, atol=2e-5, err_msg=name)


@pytest.mark.parametrize('S,F', [(48, 48), (96, 96), (20, 37), (128, 128)])
def test_render_fine_samples_on_bin_edges(dev, S, F):
    """The merge places a fine sample from the bin it was drawn in (three comparisons with the neighbouring coarse depths).  Draws
    of exactly 0 and of 1 - 2^-24 put fine samples on the first midpoint and at the very end of the last bin, repeated draws put
    many into one bin, and jitter of 0 / ~1 makes neighbouring coarse depths (nearly) coincide with the midpoints between them:
    sorted depths must be non-decreasing, every slot written, and equal to the reference's sort."""
    import gnerf_hip
    from oracle import render_ref as R
    planes, dec, o, d, nc, nf = _random_scene(31, N=1, res=8, S=S, F=F, hw=(24, 24))
    nf, nc = nf.clone(), nc.clone()
    nf[:, 0] = 0.0
    nf[:, 1] = 1.0 - 2.0 ** -24
    nf[:, 2:6] = nf[:, 6:7]                     # five samples from one draw
    nf[5, :] = torch.linspace(0, 1 - 2.0 ** -24, F)
    nf[6, :] = 0.0
    nf[7, :] = 1.0 - 2.0 ** -24
    nc[:, 8:16, 0::3] = 0.0
    nc[:, 8:16, 1::3] = 1.0 - 2.0 ** -24
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus')
    st = {}
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf, stages=st)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    rgb, depth, wsum, dbg = gnerf_hip.render_forward(nhwc, 1, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev),
                                                     depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                                                     image_width=8, debug=True)
    sorted_d = dbg[:, 5].cpu()
    assert bool((sorted_d[:, 1:] >= sorted_d[:, :-1]).all())
    np.testing.assert_allclose(sorted_d.numpy(), st['depths_all'].numpy(), atol=5e-6)
    assert float(((rgb.cpu() - ref_rgb) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(wsum.cpu().numpy(), ref_w.numpy(), atol=2e-4)


@pytest.mark.parametrize('launcher', ['torch.distributed.run', 'self'])
def test_bench_two_ranks_on_one_gpu(dev, launcher):
    """The multi-rank path of bench.py end to end (rendezvous, per-rank scene, barriers, max-over-ranks timing, one JSON line from
    rank 0), rehearsed with two processes sharing this box's one GPU over gloo (GNERF_DIST_BACKEND; the driver's 8-GPU run uses
    RCCL).  Not a scaling number.  launcher 'self': plain `python bench.py --gpus 2`, no torch.distributed.run around it -- the
    parent (no GPU call) starts the ranks itself (bench.self_launch)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GNERF_DIST_BACKEND='gloo', OMP_NUM_THREADS='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    head = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
            '--master-port', str(port)] if launcher != 'self' else [sys.executable]
    r = subprocess.run(head + [os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                               '--no-cpu-baseline', '--no-secondary', '--no-backward'], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # library chatter (gloo / RCCL banners) must not reach stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 5 and line['scaling'] == 'weak' and line['value'] > 1e6
    assert line['ranks_seen'] == 2 and line['backend'] == 'gloo' and line['self_launched'] == (launcher == 'self')
    assert [x['rank'] for x in line['ranks']] == [0, 1] and all(x['device'].startswith('cuda') for x in line['ranks'])
    assert line['roofline']['kernel_ms'] > 0 and line['cpu_baseline'] is None and line['secondary'] is None
    # per-rank values (so that a scaling run explains itself) and both plane layouts at top level
    assert [r['rank'] for r in line['per_rank']] == [0, 1] and all(r['value'] > 5e5 and r['render_call_ms'] > 0 for r in line['per_rank'])
    assert line['producer_layout_step']['value'] > line['value'] * 0.9 and line['config']['producer_layout_value'] == line['producer_layout_step']['value']


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_channels_last_forms_equal_nchw_forms(dev, dtype):
    """The channels_last kernels (csrc/modconv.hip *_nhwc, the channels_last 4x4 blur of csrc/upfirdn2d.hip) against the NCHW
    kernels on the same values: scale_channels and the epilogue are bit-identical (same roundings), the epilogue with next_scale
    equals epilogue followed by scale_channels, the blur agrees with the PyTorch-op form to the storage type's rounding; every
    result keeps the input's memory format."""
    import gnerf_hip
    from torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(3)
    half = dtype == torch.float16
    for (n, c, h, w) in [(3, 16, 12, 20), (1, 64, 33, 17), (2, 8, 5, 7)]:
        x = (torch.randn(n, c, h, w, generator=gen) * 3).to(dev).to(dtype)
        xc = x.contiguous(memory_format=torch.channels_last)
        assert gnerf_hip.is_channels_last(xc)
        sc = (torch.randn(n, c, generator=gen) + 1).to(dev)
        nx = (torch.randn(n, c, generator=gen) + 1).to(dev)
        b = torch.randn(c, generator=gen).to(dev)
        got = gnerf_hip.scale_channels(xc, sc)
        assert gnerf_hip.is_channels_last(got) and torch.equal(got, gnerf_hip.scale_channels(x, sc))
        for noise in (None, torch.randn(h, w, generator=gen).to(dev), torch.randn(n, 1, h, w, generator=gen).to(dev)):
            for scale in (None, sc):
                for rn in (False, True):
                    kw = dict(scale=scale, noise=noise, round_noise=rn, act='lrelu', gain=1.3, clamp=2.5)
                    want = gnerf_hip.modconv_epilogue(x, b, **kw)
                    got = gnerf_hip.modconv_epilogue(xc, b, **kw)
                    assert gnerf_hip.is_channels_last(got) and torch.equal(got, want), (n, c, noise is None, scale is None, rn)
                    both = gnerf_hip.modconv_epilogue(xc, b, next_scale=nx, **kw)
                    assert torch.equal(both, gnerf_hip.scale_channels(got, nx))
        assert torch.equal(gnerf_hip.modconv_epilogue(xc, b, act='linear', clamp=2.0), gnerf_hip.modconv_epilogue(x, b, act='linear', clamp=2.0))
        with pytest.raises(RuntimeError):
            gnerf_hip.modconv_epilogue(x, b, next_scale=nx)                      # folding needs the channels_last form
        # the blur after a transposed convolution: [.., 2h+1, 2w+1] -> [.., 2h, 2w] with gain 4, and a plain filter2d
        f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
        for pad, gain, flip in (([1, 1, 1, 1], 4.0, False), ([2, 1, 2, 1], 1.0, True), ([0, 0, 0, 0], 1.0, False), ([3, 3, 3, 3], 2.0, False)):
            if h + pad[2] + pad[3] < 4 or w + pad[0] + pad[1] < 4:
                continue
            got = upfirdn2d.upfirdn2d(xc, f, padding=pad, gain=gain, flip_filter=flip)
            want = upfirdn2d.upfirdn2d(x.float(), f, padding=pad, gain=gain, flip_filter=flip, impl='ref')
            assert got.dtype == dtype and gnerf_hip.is_channels_last(got) and got.shape == want.shape
            tol = dict(rtol=2e-3, atol=2e-3 * float(want.abs().max())) if half else dict(rtol=1e-5, atol=1e-5 * float(want.abs().max()))
            np.testing.assert_allclose(got.float().cpu().numpy(), want.cpu().numpy(), **tol)
        g = torch.randn(4, 4, generator=gen).to(dev)                             # a non-symmetric filter: flip and orientation matter
        got = upfirdn2d.upfirdn2d(xc, g, padding=[1, 2, 2, 1])
        want = upfirdn2d.upfirdn2d(x.float(), g, padding=[1, 2, 2, 1], impl='ref')
        np.testing.assert_allclose(got.float().cpu().numpy(), want.cpu().numpy(), rtol=2e-3 if half else 1e-5, atol=(4e-3 if half else 1e-5) * float(want.abs().max()))


@pytest.mark.parametrize('dtype', [torch.float16, torch.float32])
def test_blur_epilogue_fused_equals_two_passes(dev, dtype):
    """gnerf_blur4_epilogue_nhwc against gnerf_upfirdn2d followed by gnerf_modconv_epilogue_nhwc on channels_last activations: bit for
    bit (the blurred value is rounded to the storage type in between, as the materialised tensor would be)."""
    import gnerf_hip
    from torch_utils.ops import upfirdn2d
    gen = torch.Generator().manual_seed(9)
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    g = torch.randn(4, 4, generator=gen).to(dev)
    for (n, c, h, w) in [(3, 16, 13, 21), (1, 64, 33, 17), (2, 8, 5, 7), (1, 128, 65, 65)]:
        x = (torch.randn(n, c, h, w, generator=gen) * 3).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
        sc = (torch.randn(n, c, generator=gen) + 1).to(dev)
        nx = (torch.randn(n, c, generator=gen) + 1).to(dev)
        b = torch.randn(c, generator=gen).to(dev)
        for filt, pad, bg, flip in ((f, [1, 1, 1, 1], 4.0, False), (g, [2, 1, 1, 2], 1.0, True), (f, [0, 0, 0, 0], 1.0, False)):
            if h + pad[2] + pad[3] < 4 or w + pad[0] + pad[1] < 4:
                continue
            blurred = upfirdn2d.upfirdn2d(x, filt, padding=pad, gain=bg, flip_filter=flip)
            for scale in (None, sc):
                for nxt in (None, nx):
                    for act, clamp in (('lrelu', 2.5), ('linear', None)):
                        kw = dict(bias=b, scale=scale, act=act, gain=1.3, clamp=clamp, next_scale=nxt)
                        want = gnerf_hip.modconv_epilogue(blurred, **kw)
                        got = gnerf_hip.blur_epilogue_channels_last(x, filt, pad, blur_gain=bg, flip_filter=flip, **kw)
                        assert gnerf_hip.is_channels_last(got) and got.shape == want.shape
                        assert torch.equal(got, want), (n, c, pad, scale is None, nxt is None, act)
    with pytest.raises(RuntimeError):
        gnerf_hip.blur_epilogue_channels_last(x.contiguous(), f, [1, 1, 1, 1])
    if dtype == torch.float16:
        # one large image (an orbit frame's layers): the strip length follows the launch's parallelism (8 .. 64 rows), and the result
        # must not -- against the PyTorch-op blur (upfirdn2d.py:168-213) + the separate epilogue, to the storage type's rounding
        for (n, c, h) in [(1, 64, 513), (1, 32, 131), (2, 128, 257)]:
            x = (torch.randn(n, c, h, h, generator=gen) * 3).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
            sc, nx, b = (torch.randn(n, c, generator=gen) + 1).to(dev), (torch.randn(n, c, generator=gen) + 1).to(dev), torch.randn(c, generator=gen).to(dev)
            kw = dict(bias=b, scale=sc, act='lrelu', gain=1.3, clamp=256.0, next_scale=nx)
            got = gnerf_hip.blur_epilogue_channels_last(x, f, [1, 1, 1, 1], blur_gain=4.0, **kw)
            two = gnerf_hip.modconv_epilogue(upfirdn2d.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4.0), **kw)
            assert torch.equal(got, two)
            ref = upfirdn2d.upfirdn2d(x.float(), f, padding=[1, 1, 1, 1], gain=4.0, impl='ref').to(dtype).contiguous(memory_format=torch.channels_last)
            want = gnerf_hip.modconv_epilogue(ref, **kw)
            diff = (got.float() - want.float()).abs()
            assert float(diff.max()) <= 2e-2 * float(want.float().abs().max()) and float(diff.mean()) <= 2e-4 * float(want.float().abs().max())


def test_torgb_channels_last_vs_composed_ops(dev):
    """gnerf_torgb_nhwc against ToRGBLayer's op chain (networks_stylegan2.py:349-367, fused modulation :89-96): fp16 modulated
    weights, fp32 accumulation, bias, clamp.  The kernel rounds once where the chain rounds the convolution's output and the bias
    sum separately, so the comparison allows two fp16 roundings."""
    import gnerf_hip
    gen = torch.Generator().manual_seed(5)
    for (n, c, h, w) in [(4, 128, 24, 40), (1, 256, 16, 16), (2, 32, 9, 13), (1, 512, 4, 4), (3, 64, 7, 5)]:
        x = (torch.randn(n, c, h, w, generator=gen) * 2).to(dev).half().contiguous(memory_format=torch.channels_last)
        weight = torch.randn(3, c, 1, 1, generator=gen).to(dev)
        styles = ((torch.randn(n, c, generator=gen) + 1) / math.sqrt(c)).to(dev)
        bias = torch.randn(3, generator=gen).to(dev)
        for clamp in (None, 0.75):
            got = gnerf_hip.torgb_channels_last(x, weight, styles, bias, clamp=clamp)
            assert got.shape == (n, 3, h, w) and got.dtype == torch.float16 and got.is_contiguous()
            wmod = (weight.reshape(1, 3, c) * styles[:, None, :]).half().float()                     # [n,3,c]
            acc = torch.einsum('nchw,noc->nohw', x.float(), wmod).half().float() + bias.half().float()[None, :, None, None]
            want = acc if clamp is None else acc.clamp(-clamp, clamp)
            np.testing.assert_allclose(got.float().cpu().numpy(), want.half().float().cpu().numpy(), rtol=2e-3, atol=2e-3 * float(want.abs().max()))
    with pytest.raises(RuntimeError):
        gnerf_hip.torgb_channels_last(x.contiguous(), weight, styles, bias)
    # accumulate form: the layer added to the block's running fp32 image in the same launch == img.add_(y.to(float32)), bit for bit
    for (n, c, h, w) in [(4, 128, 24, 40), (1, 64, 33, 17), (2, 32, 9, 13)]:
        x = (torch.randn(n, c, h, w, generator=gen) * 2).to(dev).half().contiguous(memory_format=torch.channels_last)
        weight = torch.randn(3, c, 1, 1, generator=gen).to(dev)
        styles = ((torch.randn(n, c, generator=gen) + 1) / math.sqrt(c)).to(dev)
        bias = torch.randn(3, generator=gen).to(dev)
        img = torch.randn(n, 3, h, w, generator=gen).to(dev)
        for clamp in (None, 0.75):
            want = img + gnerf_hip.torgb_channels_last(x, weight, styles, bias, clamp=clamp).float()
            acc = img.clone()
            got = gnerf_hip.torgb_channels_last(x, weight, styles, bias, clamp=clamp, accumulate_into=acc)
            assert got is acc and torch.equal(got, want)
        with pytest.raises(RuntimeError):
            gnerf_hip.torgb_channels_last(x, weight, styles, bias, accumulate_into=img.half())
        with pytest.raises(RuntimeError):
            gnerf_hip.torgb_channels_last(x, weight, styles, bias, accumulate_into=img[:, :, :, ::2])
