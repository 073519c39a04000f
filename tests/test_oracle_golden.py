"""Pin the CPU oracle (oracle/) to the golden vectors captured from the reference.

CPU only.  If these pass, the oracle is a faithful restatement of the reference's
PyTorch path on the recorded inputs, and the GPU parity tests may use it as the checker.
"""

import numpy as np
import pytest
import torch

from oracle import render_ref as R
from oracle import ops_ref as O

RENDER_CASES = ['render_s12.npz', 'render_s48.npz', 'render_misc.npz', 'render_nofine.npz', 'render_dnoise.npz']       # (the last: density_noise = 0.5, renderer.py:146-147)


def _t(a, dt=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dt)


def _options(g):
    return dict(depth_resolution=int(g['depth_resolution']), depth_resolution_importance=int(g['depth_resolution_importance']),
                ray_start=float(g['ray_start']), ray_end=float(g['ray_end']), box_warp=float(g['box_warp']),
                clamp_mode='softplus', white_back=bool(g['white_back']), disparity_space_sampling=bool(g['disparity']))


def _run(g, dt):
    dec = R.fold_decoder(_t(g['w1'], dt), _t(g['b1'], dt), _t(g['w2'], dt), _t(g['b2'], dt), float(g['lr_mul']))
    stages = {}
    noise_f = _t(g['noise_fine'], dt) if 'noise_fine' in g else None
    sn = (_t(g['sigma_noise_coarse'], dt), _t(g['sigma_noise_fine'], dt)) if 'sigma_noise_coarse' in g else None
    out = R.render(_t(g['planes'], dt), dec, _t(g['ray_origins'], dt), _t(g['ray_dirs'], dt), _options(g),
                   _t(g['noise_coarse'], dt), noise_f, stages, sigma_noise=sn)
    return out, stages


@pytest.mark.parametrize('case', RENDER_CASES)
def test_render_matches_reference(golden, case):
    g = golden(case)
    (rgb, depth, wsum), st = _run(g, torch.float32)
    N, M = g['out_rgb'].shape[:2]
    # depth proposals are pure arithmetic on recorded noise: expect (near) bit-exactness
    np.testing.assert_allclose(st['depths_coarse'].reshape(N, M, -1).numpy(), g['depths_coarse'], rtol=0, atol=2.4e-7)
    np.testing.assert_allclose(st['sigma_coarse'].reshape(N, M, -1).numpy(), g['sigma_coarse'], rtol=1e-4, atol=2e-5)
    if 'depths_fine' in g:
        np.testing.assert_allclose(st['weights_coarse'].reshape(N, M, -1).numpy(), g['weights_coarse'], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(st['depths_fine'].reshape(N, M, -1).numpy(), g['depths_fine'], rtol=0, atol=2e-5)
        np.testing.assert_allclose(st['depths_all'].reshape(N, M, -1).numpy(), g['depths_all'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(rgb.numpy(), g['out_rgb'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(depth.numpy(), g['out_depth'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(wsum.numpy(), g['out_wsum'], rtol=0, atol=2e-5)
    mse = float(((rgb.numpy() - g['out_rgb']) ** 2).mean())
    assert mse < 1e-9            # north_star bound is 1e-4; the oracle sits 5 orders below it


@pytest.mark.parametrize('case', RENDER_CASES)
def test_render_fp64_truth_is_close(golden, case):
    """The float64 run of the oracle is the tie-breaker 'truth'; it must agree with the fp32 reference to fp32 noise."""
    g = golden(case)
    (rgb, depth, wsum), _ = _run(g, torch.float64)
    assert float(((rgb.numpy() - g['out_rgb']) ** 2).mean()) < 1e-8
    np.testing.assert_allclose(wsum.numpy(), g['out_wsum'], atol=1e-4)


def test_make_rays(golden):
    g = golden('camera.npz')
    o, d = R.make_rays(_t(g['rs_cam2world']), _t(g['rs_intrinsics']), int(g['rs_res']))
    np.testing.assert_allclose(o.numpy(), g['rs_origins'], rtol=0, atol=0)
    np.testing.assert_allclose(d.numpy(), g['rs_dirs'], rtol=0, atol=1.2e-7)
    for case in RENDER_CASES[:2]:
        gg = golden(case)
        o, d = R.make_rays(_t(gg['cam2world']), _t(gg['intrinsics']), int(gg['res']))
        np.testing.assert_allclose(o.numpy(), gg['ray_origins'], atol=0)
        np.testing.assert_allclose(d.numpy(), gg['ray_dirs'], atol=1.2e-7)


def test_orbit_cameras(golden):
    g = golden('camera.npz')
    for i, ref in zip(g['orbit_frames'], g['orbit_cam2world']):
        yaw = 3.14 / 2 + 0.7 * np.sin(2 * 3.14 * i / 120)
        pitch = 3.14 / 2 - 0.05 + 0.3 * np.cos(2 * 3.14 * i / 120)
        m = R.lookat_pose(yaw, pitch, 2.7)
        np.testing.assert_allclose(m[0].numpy(), ref, atol=3e-7)


def test_importance_depths_edge_cases(golden):
    g = golden('stages.npz')
    out = R.importance_depths(_t(g['pdf_depths']), _t(g['pdf_weights']), _t(g['pdf_noise']))
    np.testing.assert_allclose(out.numpy(), g['pdf_out'], rtol=0, atol=3e-6)


def test_composite_edge_cases(golden):
    g = golden('stages.npz')
    col, sig, dep = _t(g['march_colors'])[0], _t(g['march_sigma'])[0, :, :, 0], _t(g['march_depths'])[0, :, :, 0]
    for wb in (0, 1):
        rgb, depth, w = R.composite(col, sig, dep, bool(wb), (dep.min(), dep.max()))
        np.testing.assert_allclose(w.numpy(), g[f'march_w_wb{wb}'][0, :, :, 0], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(rgb.numpy(), g[f'march_rgb_wb{wb}'][0], rtol=0, atol=1e-6)
        np.testing.assert_allclose(depth.numpy(), g[f'march_depth_wb{wb}'][0, :, 0], rtol=0, atol=1e-6)
    # the underflow ray: zero total weight -> NaN -> +inf -> clamped to the GLOBAL max depth (ray_marcher.py:49-50)
    assert g['march_w_wb0'][0, 1].sum() == 0.0
    assert g['march_depth_wb0'][0, 1, 0] == g['march_depths'].max()


def test_plane_projection_and_lookup(golden):
    g = golden('stages.npz')
    pts = _t(g['proj_points'])[0]
    uv = R.plane_uv(pts, 1.0)
    np.testing.assert_allclose(uv.numpy(), g['proj_uv'], atol=0)
    planes = _t(g['lookup_planes'])[0]
    for p in range(3):
        f = R.bilinear_zeros(planes[p], uv[p])
        np.testing.assert_allclose(f.numpy(), g['lookup_out'][0, p], rtol=0, atol=1e-6)
    assert (np.abs(g['proj_uv']) > 1).any()      # the fixture does exercise zero padding


# ----------------------------------------------------------------------------
# ops


@pytest.mark.parametrize('act', list(O.ACTIVATIONS))
@pytest.mark.parametrize('clamp', [None, 0.9])
def test_bias_act_all_orders(golden, act, clamp):
    g = golden('ops.npz')
    tag = f'ba_{act}_{"c" if clamp else "n"}'
    x, b, dy, ddx = g['ba_x'], g['ba_b'], g['ba_dy'], g['ba_ddx']
    np.testing.assert_allclose(O.bias_act(x, b, 1, act, clamp=clamp), g[tag + '_y'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.bias_act_grad(dy, x, b, 1, act, clamp=clamp), g[tag + '_dx'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(O.bias_act_grad2(ddx, dy, x, b, 1, act, clamp=clamp), g[tag + '_d2'], rtol=1e-9, atol=1e-12)


def test_bias_act_dim0(golden):
    g = golden('ops.npz')
    y = O.bias_act(g['ba2_x'], g['ba2_b'], dim=0, act='lrelu', alpha=0.1, gain=0.7)
    np.testing.assert_allclose(y, g['ba2_y'], rtol=1e-6, atol=1e-7)


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
def test_upfirdn2d(golden, name):
    g = golden('ops.npz')
    kw = dict(UP_CASES[name])
    f = kw.pop('f')
    f = None if f is None else g['up_' + f]
    y = O.upfirdn2d(g['up_x'], f, **kw)
    assert y.shape == g['up_' + name].shape
    np.testing.assert_allclose(y, g['up_' + name], rtol=1e-5, atol=2e-6)


def test_setup_filter(golden):
    g = golden('ops.npz')
    np.testing.assert_allclose(O.setup_filter([1, 3, 3, 1]), g['up_f4'], rtol=1e-7)
    np.testing.assert_allclose(O.setup_filter([1., 2., 3., 4., 3., 2., 1., 0.5]), g['up_fs'], rtol=1e-7)


def test_filtered_lrelu(golden):
    g = golden('ops.npz')
    x, b, fu, fd = g['fl_x'], g['fl_b'], g['fl_fu'], g['fl_fd']
    y = O.filtered_lrelu(x, fu, fd, b, up=2, down=2, padding=[10, 10, 10, 10], gain=1.3, slope=0.1, clamp=0.8)
    np.testing.assert_allclose(y, g['fl_up2_down2'], rtol=1e-4, atol=2e-6)
    y = O.filtered_lrelu(x, fu, fd, b, up=4, down=2, padding=[11, 10, 9, 12], flip_filter=True)
    np.testing.assert_allclose(y, g['fl_up4_down2'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(O.filtered_lrelu(x, b=b), g['fl_plain'], rtol=1e-5, atol=1e-6)


def test_grid_sample_oracle_matches_torch():
    """The numpy restatement of the sampler against the op the reference calls (torch's, on the CPU), incl. out-of-range points."""
    import torch
    from oracle import ops_ref as O
    g = torch.Generator().manual_seed(0)
    img = torch.randn(2, 3, 7, 9, generator=g)
    grid = torch.rand(2, 5, 6, 2, generator=g) * 2.6 - 1.3
    want = torch.nn.functional.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False).numpy()
    np.testing.assert_allclose(O.grid_sample_2d(img.numpy(), grid.numpy()), want, rtol=1e-5, atol=1e-6)


def test_philox_known_answers():
    """oracle/philox_ref.philox4x32_10 against the Random123 known-answer vectors of Philox4x32-10 (Salmon et al., kat_vectors:
    counter / key all zero, all ones, and the digits of pi) -- the generator under torch's device uniform draws."""
    from oracle import philox_ref as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = P.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    # the uniform mapping: (0, 1] from the bits, 1 sent back to 0; never negative, never 1
    u = P.uniform_from_bits(np.array([0, 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32 - 200], dtype=np.uint32))
    assert u[0] == np.float32(2.0 ** -32) and u[3] == 0.0 and 0.0 <= u.min() and u.max() < 1.0


def test_torch_rand_plan_matches_restatement():
    """gnerf_torch_rand_plan (host arithmetic of the C ABI, no GPU) == oracle/philox_ref's launch geometry of ATen's uniform kernel."""
    import ctypes
    import gnerf_hip
    from oracle import philox_ref as P
    lib = gnerf_hip.load()
    for numel in (1, 255, 256, 257, 4096, 2 ** 19, 2 ** 19 + 1, 4 * 16384 * 48, 65536 * 96, 10 ** 8):
        for mp, mt in ((256, 2048), (304, 2048), (120, 2560), (1, 256)):
            thr, inc = ctypes.c_uint32(0), ctypes.c_uint64(0)
            assert lib.gnerf_torch_rand_plan(numel, mp, mt, ctypes.byref(thr), ctypes.byref(inc)) == 0
            g = P.grid_threads(numel, mp, mt)
            assert (thr.value, inc.value) == (g, P.offset_increment(numel, g)), (numel, mp, mt)
