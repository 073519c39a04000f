"""CPU tests: the C-ABI library loads and exports what include/gnerf_hip.h declares; the drop-in
Python modules (PyTorch-op paths, i.e. what the reference does for CPU tensors) reproduce the golden
vectors; the overlay packages resolve the way the reference's imports need."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'gnerf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(gnerf_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import gnerf_hip
    names = _declared_functions()
    assert 'gnerf_render_forward' in names and 'gnerf_bias_act' in names and 'gnerf_upfirdn2d' in names
    lib = gnerf_hip.load()          # raises if the .so is missing: there is no fallback
    for n in names:
        assert hasattr(lib, n), f'{n} declared in gnerf_hip.h but not exported'
        assert n in gnerf_hip.SIGNATURES, f'{n} has no ctypes signature'
    assert sorted(gnerf_hip.SIGNATURES) == names
    assert lib.gnerf_abi_version() == gnerf_hip.ABI_VERSION
    assert b'gfx950' in lib.gnerf_build_info()


@pytest.mark.parametrize('struct,mirror', [('gnerf_render_params', 'RenderParams'), ('gnerf_render_grads', 'RenderGrads')])
def test_render_structs_match_header(struct, mirror):
    """Field order of each ctypes mirror == field order of the struct in the header."""
    import gnerf_hip
    text = open(os.path.join(ROOT, 'include', 'gnerf_hip.h')).read()
    body = text[text.index('typedef struct %s {' % struct):text.index('} %s;' % struct)]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for stmt in body.split('{', 1)[1].split(';'):
        stmt = stmt.strip()
        if not stmt:
            continue
        for part in stmt.split(','):
            fields.append(re.findall(r'[A-Za-z_0-9]+', part)[-1])
    assert fields == [f[0] for f in getattr(gnerf_hip, mirror)._fields_]


def test_torch_extension_binding_exports_the_reference_plugin_entry_points():
    """gnerf_torch_ext.so (csrc/torch_binding.cpp) loads next to libgnerf_hip.so and exports pybind functions with the argument
    lists of the reference's plugins (bias_act.cpp:36, upfirdn2d.cpp:20, filtered_lrelu.cpp:20, :217); get_plugin hands them out.
    No compute here (no GPU): CPU tensors are refused by every entry point."""
    import gnerf_hip
    from torch_utils import custom_ops
    e = gnerf_hip.ext()
    assert e is not None, 'gnerf_torch_ext.so is not built (g-nerf_amd/csrc/build.sh)'
    assert e.abi_version() == gnerf_hip.ABI_VERSION
    want = {
        'bias_act': '(arg0: torch.Tensor, arg1: torch.Tensor, arg2: torch.Tensor, arg3: torch.Tensor, arg4: torch.Tensor, arg5: typing.SupportsInt, '
                    'arg6: typing.SupportsInt, arg7: typing.SupportsInt, arg8: typing.SupportsFloat, arg9: typing.SupportsFloat, arg10: typing.SupportsFloat) -> torch.Tensor',
        'upfirdn2d': 12, 'filtered_lrelu': 18, 'filtered_lrelu_act_': 8,
    }
    for name, sig in want.items():
        doc = getattr(e, name).__doc__
        n_args = doc.split(' -> ')[0].count('arg')
        assert n_args == (sig if isinstance(sig, int) else 11), (name, doc)
    assert 'SupportsInt' in e.bias_act.__doc__ or 'int' in e.bias_act.__doc__
    custom_ops._cached_plugins.clear()
    old = custom_ops.verbosity
    custom_ops.verbosity = 'none'
    try:
        assert custom_ops.get_plugin('bias_act_plugin', sources=[]).bias_act is e.bias_act
        assert custom_ops.get_plugin('upfirdn2d_plugin', sources=[]).upfirdn2d is e.upfirdn2d
        fl = custom_ops.get_plugin('filtered_lrelu_plugin', sources=[])
        assert fl.filtered_lrelu is e.filtered_lrelu and fl.filtered_lrelu_act_ is e.filtered_lrelu_act_
    finally:
        custom_ops.verbosity = old
    x, null = torch.zeros(2, 3, 4, 4), torch.empty([0])
    with pytest.raises(RuntimeError, match='CUDA device'):
        e.bias_act(x, null, null, null, null, 0, 1, 1, 0.0, 1.0, -1.0)
    with pytest.raises(RuntimeError, match='CUDA device'):
        e.upfirdn2d(x, torch.ones(2, 2), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)


def test_gpu_ops_refuse_cpu_tensors():
    """The native entry points never compute on the host."""
    import gnerf_hip
    x = torch.zeros(4, 4)
    with pytest.raises(RuntimeError):
        gnerf_hip.bias_act(x, None, None, None, None, 0, 1, 1, 0.0, 1.0, -1.0)
    with pytest.raises(RuntimeError):
        gnerf_hip.planes_to_nhwc(torch.zeros(3, 32, 4, 4))
    with pytest.raises(RuntimeError):
        gnerf_hip.modconv_epilogue(torch.zeros(1, 8, 4, 4), torch.zeros(8))
    with pytest.raises(RuntimeError):
        gnerf_hip.torgb_channels_last(torch.zeros(1, 32, 4, 4, dtype=torch.float16).contiguous(memory_format=torch.channels_last),
                                      torch.zeros(3, 32), torch.zeros(1, 32))


def test_activation_layout_detection():
    """Host logic that routes activations to the NCHW or the channels_last kernels (gnerf_hip._activation_layout)."""
    import gnerf_hip
    x = torch.zeros(2, 8, 4, 6)
    assert gnerf_hip._activation_layout(x, 't') == 'nchw' and not gnerf_hip.is_channels_last(x)
    xc = x.contiguous(memory_format=torch.channels_last)
    assert gnerf_hip._activation_layout(xc, 't') == 'nhwc' and gnerf_hip.is_channels_last(xc)
    one = torch.zeros(2, 1, 4, 6).contiguous(memory_format=torch.channels_last)          # C == 1: both formats describe the same memory
    assert gnerf_hip._activation_layout(one, 't') == 'nchw' and not gnerf_hip.is_channels_last(one)
    with pytest.raises(RuntimeError):
        gnerf_hip._activation_layout(x[:, :, :, ::2], 't')                                # neither dense format
    with pytest.raises(RuntimeError):
        gnerf_hip._activation_layout(torch.zeros(2, 8, 4), 't')
    with pytest.raises(RuntimeError):
        gnerf_hip._activation_layout(torch.zeros(2, 8, 4, 6, dtype=torch.float64), 't')


# ---------------------------------------------------------------------------- drop-in modules on CPU


class _Replay:
    """Feed recorded noise to torch.rand_like / torch.rand (the reference's two draws)."""

    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self._rl, self._r = torch.rand_like, torch.rand
        torch.rand_like = lambda t, **k: self.draws.pop(0).reshape(t.shape).to(t.dtype)
        torch.rand = lambda *a, **k: self.draws.pop(0)
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.rand = self._rl, self._r


class _FC(torch.nn.Module):
    """FullyConnectedLayer stand-in with the attributes the renderer inspects (networks_stylegan2.py:101-134)."""

    def __init__(self, w, b, lr_mul):
        super().__init__()
        self.weight = torch.nn.Parameter(w, requires_grad=False)
        self.bias = torch.nn.Parameter(b, requires_grad=False)
        self.activation = 'linear'
        self.weight_gain = lr_mul / np.sqrt(w.shape[1])
        self.bias_gain = lr_mul

    def forward(self, x):
        return torch.addmm((self.bias * self.bias_gain).unsqueeze(0), x, (self.weight * self.weight_gain).t())


class Decoder(torch.nn.Module):
    """OSGDecoder stand-in (triplane.py:113-136)."""

    def __init__(self, g):
        super().__init__()
        lr = float(g['lr_mul'])
        self.net = torch.nn.Sequential(_FC(torch.from_numpy(g['w1']), torch.from_numpy(g['b1']), lr), torch.nn.Softplus(),
                                       _FC(torch.from_numpy(g['w2']), torch.from_numpy(g['b2']), lr))

    def forward(self, feats, dirs):
        x = feats.mean(1)
        N, M, C = x.shape
        x = self.net(x.view(N * M, C)).view(N, M, -1)
        return {'rgb': torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001, 'sigma': x[..., 0:1]}


def options_of(g):
    return dict(depth_resolution=int(g['depth_resolution']), depth_resolution_importance=int(g['depth_resolution_importance']),
                ray_start=float(g['ray_start']), ray_end=float(g['ray_end']), box_warp=float(g['box_warp']), clamp_mode='softplus',
                white_back=bool(g['white_back']), disparity_space_sampling=bool(g['disparity']),
                superresolution_module='ignored', c_gen_conditioning_zero=True)     # unknown keys must be ignored


@pytest.mark.parametrize('case', ['render_s12.npz', 'render_s48.npz', 'render_misc.npz', 'render_nofine.npz'])
def test_renderer_torch_path_matches_reference(golden, case):
    from training.volumetric_rendering.renderer import ImportanceRenderer
    g = golden(case)
    ren = ImportanceRenderer()
    draws = [torch.from_numpy(g['noise_coarse'])] + ([torch.from_numpy(g['noise_fine'])] if 'noise_fine' in g else [])
    with torch.no_grad(), _Replay(draws):
        rgb, depth, wsum = ren(torch.from_numpy(g['planes']), Decoder(g), torch.from_numpy(g['ray_origins']),
                               torch.from_numpy(g['ray_dirs']), options_of(g))
    np.testing.assert_allclose(rgb.numpy(), g['out_rgb'], atol=2e-6)
    np.testing.assert_allclose(depth.numpy(), g['out_depth'], atol=2e-6)
    np.testing.assert_allclose(wsum.numpy(), g['out_wsum'], atol=2e-6)


def test_renderer_views_of_one_item_on_the_torch_path():
    """The overlay's extension of ImportanceRenderer.forward -- planes of ONE item, rays of N: N views, each with the results (draws,
    depth clamp) of a call of its own -- on the PyTorch-op path, and through Generator.synthesis (several cameras, one latent)."""
    import gnerf_harness as H
    import gnerf_generator
    from training.volumetric_rendering.renderer import ImportanceRenderer
    from training.volumetric_rendering.ray_sampler import RaySampler
    torch.manual_seed(0)
    dec = H.TriPlaneDecoder()
    opts = dict(depth_resolution=6, depth_resolution_importance=5, ray_start=2.25, ray_end=3.3, box_warp=1, clamp_mode='softplus',
                disparity_space_sampling=False)
    c = torch.cat([H.camera_label(H.orbit_pose(i, 120)) for i in (0, 31, 77)])
    o, d = RaySampler()(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 6)
    planes = torch.randn(1, 3, 32, 16, 16)
    ren = ImportanceRenderer()
    with torch.no_grad():
        torch.manual_seed(4)
        got = ren(planes, dec, o, d, opts)
        torch.manual_seed(4)
        for i in range(3):
            one = ren(planes, dec, o[i:i + 1], d[i:i + 1], opts)
            for a, e in zip(got, one):
                assert torch.equal(a[i:i + 1], e)
        assert got[0].shape == (3, 36, 32) and got[1].shape == (3, 36, 1)
        G = gnerf_generator.Generator(sr_use_fp16=False).eval()
        ws = G.mapping(torch.randn(1, 512), torch.zeros(1, 25))
        G.synthesis(ws, c[:1], neural_rendering_resolution=8, noise_mode='const', cache_backbone=True)
        torch.manual_seed(5)
        together = G.synthesis(ws, c, neural_rendering_resolution=8, noise_mode='const', use_cached_backbone=True)
        assert together['image'].shape == (3, 3, 512, 512) and together['image_depth'].shape == (3, 1, 8, 8)
        torch.manual_seed(5)
        for i in range(3):
            alone = G.synthesis(ws, c[i:i + 1], neural_rendering_resolution=8, noise_mode='const', use_cached_backbone=True)
            assert torch.equal(alone['image_depth'], together['image_depth'][i:i + 1])
            assert torch.allclose(alone['image'], together['image'][i:i + 1], atol=1e-4)


def test_renderer_survives_unpickling_without_init(golden):
    """legacy.py:68-72 revives ImportanceRenderer by class name WITHOUT calling __init__: only the pickled
    attributes exist.  The replacement must cope (lazy state only)."""
    import pickle
    from training.volumetric_rendering.renderer import ImportanceRenderer
    blob = pickle.dumps(ImportanceRenderer())
    ren = pickle.loads(blob)
    assert '_gnerf_planes_cache' not in ren.__dict__
    g = golden('render_nofine.npz')
    with torch.no_grad(), _Replay([torch.from_numpy(g['noise_coarse'])]):
        rgb, _, _ = ren(torch.from_numpy(g['planes']), Decoder(g), torch.from_numpy(g['ray_origins']), torch.from_numpy(g['ray_dirs']), options_of(g))
    np.testing.assert_allclose(rgb.numpy(), g['out_rgb'], atol=2e-6)


def test_ray_sampler_and_marcher_cpu(golden):
    from training.volumetric_rendering.ray_sampler import RaySampler
    from training.volumetric_rendering.ray_marcher import MipRayMarcher2
    g = golden('camera.npz')
    o, d = RaySampler()(torch.from_numpy(g['rs_cam2world']), torch.from_numpy(g['rs_intrinsics']), int(g['rs_res']))
    np.testing.assert_allclose(o.numpy(), g['rs_origins'], atol=0)
    np.testing.assert_allclose(d.numpy(), g['rs_dirs'], atol=1e-7)
    s = golden('stages.npz')
    for wb in (0, 1):
        rgb, depth, w = MipRayMarcher2()(torch.from_numpy(s['march_colors']), torch.from_numpy(s['march_sigma']),
                                         torch.from_numpy(s['march_depths']), {'clamp_mode': 'softplus', 'white_back': bool(wb)})
        np.testing.assert_allclose(rgb.numpy(), s[f'march_rgb_wb{wb}'], atol=1e-6)
        np.testing.assert_allclose(depth.numpy(), s[f'march_depth_wb{wb}'], atol=1e-6)
        np.testing.assert_allclose(w.numpy(), s[f'march_w_wb{wb}'], atol=1e-7)


def test_ray_limits_box(golden):
    from training.volumetric_rendering import math_utils
    s = golden('stages.npz')
    lo, hi = math_utils.get_ray_limits_box(torch.from_numpy(s['box_origins']), torch.from_numpy(s['box_dirs']), box_side_length=1.0)
    np.testing.assert_allclose(lo.numpy(), s['box_tmin'], atol=0)
    np.testing.assert_allclose(hi.numpy(), s['box_tmax'], atol=0)
    assert (s['box_tmin'] == -1).any() and (s['box_tmin'] > 0).any()
    lin = math_utils.linspace(torch.tensor([1.0, 2.0]), torch.tensor([3.0, 6.0]), 5)
    np.testing.assert_allclose(lin.numpy(), np.linspace([1, 2], [3, 6], 5), atol=1e-6)


def test_ops_ref_paths_match_reference(golden):
    from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu
    g = golden('ops.npz')
    x, b = torch.from_numpy(g['ba_x']), torch.from_numpy(g['ba_b'])
    for act in bias_act.activation_funcs:
        y = bias_act.bias_act(x, b, act=act, clamp=0.9)      # CPU tensor -> PyTorch ops, like the reference
        np.testing.assert_allclose(y.numpy(), g[f'ba_{act}_c_y'], rtol=1e-12, atol=1e-12)
    xu = torch.from_numpy(g['up_x'])
    f4 = upfirdn2d.setup_filter([1, 3, 3, 1])
    np.testing.assert_allclose(f4.numpy(), g['up_f4'], rtol=1e-7)
    np.testing.assert_allclose(upfirdn2d.upfirdn2d(xu, f4, padding=[1, 1, 1, 1], gain=4).numpy(), g['up_blur'], atol=1e-6)
    np.testing.assert_allclose(upfirdn2d.upsample2d(xu, f4).numpy(), g['up_upsample2d'], atol=1e-6)
    np.testing.assert_allclose(upfirdn2d.downsample2d(xu, f4).numpy(), g['up_downsample2d'], atol=1e-6)
    np.testing.assert_allclose(upfirdn2d.filter2d(xu, f4).numpy(), g['up_filter2d'], atol=1e-6)
    fs = torch.from_numpy(g['up_fs'])
    np.testing.assert_allclose(upfirdn2d.upfirdn2d(xu, fs, up=2, down=3, padding=[4, 3, 5, 2], gain=2.0).numpy(), g['up_sep'], atol=2e-6)
    xf, bf = torch.from_numpy(g['fl_x']), torch.from_numpy(g['fl_b'])
    fu, fd = torch.from_numpy(g['fl_fu']), torch.from_numpy(g['fl_fd'])
    y = filtered_lrelu.filtered_lrelu(xf, fu=fu, fd=fd, b=bf, up=2, down=2, padding=[10, 10, 10, 10], gain=1.3, slope=0.1, clamp=0.8)
    np.testing.assert_allclose(y.numpy(), g['fl_up2_down2'], atol=2e-6)
    assert upfirdn2d._parse_padding(3) == (3, 3, 3, 3) and upfirdn2d._get_filter_size(f4) == (4, 4)


def test_grid_sample_gradfix_double_backward():
    from torch_utils.ops import grid_sample_gradfix
    torch.manual_seed(0)
    img = torch.randn(1, 2, 5, 6, dtype=torch.float64, requires_grad=True)
    grid = (torch.rand(1, 3, 4, 2, dtype=torch.float64) * 2.4 - 1.2)
    ref = torch.nn.functional.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=False)
    grid_sample_gradfix.enabled = True
    try:
        out = grid_sample_gradfix.grid_sample(img, grid)
        np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), atol=1e-12)
        gout = torch.randn_like(out).requires_grad_(True)
        (gi,) = torch.autograd.grad(out, img, gout, create_graph=True)
        (gg,) = torch.autograd.grad(gi.square().sum(), gout)          # second order w.r.t. grad_output
        (gi_ref,) = torch.autograd.grad(ref, img, gout.detach())
        np.testing.assert_allclose(gi.detach().numpy(), gi_ref.numpy(), atol=1e-12)
        assert gg.abs().sum() > 0
    finally:
        grid_sample_gradfix.enabled = False


def test_custom_ops_plugin_surface():
    from torch_utils import custom_ops
    custom_ops.verbosity = 'none'
    p = custom_ops.get_plugin('bias_act_plugin', sources=['x.cpp'], headers=['x.h'], source_dir='.', extra_cuda_cflags=['--use_fast_math'])
    assert callable(p.bias_act)
    assert callable(custom_ops.get_plugin('upfirdn2d_plugin', sources=[]).upfirdn2d)
    fl = custom_ops.get_plugin('filtered_lrelu_plugin', sources=[])
    assert callable(fl.filtered_lrelu) and callable(fl.filtered_lrelu_act_)
    with pytest.raises(RuntimeError):
        custom_ops.get_plugin('no_such_plugin', sources=[])


@pytest.mark.skipif(not os.path.isdir('/root/reference/g_nerf'), reason='reference tree only exists in the build container')
def test_overlay_resolves_reference_modules():
    """With g-nerf_amd/ BEFORE the reference's g_nerf/ on sys.path, hot-path modules come from this repo
    and everything else (persistence, misc, conv2d_gradfix, triplane ...) from the reference -- the
    drop-in arrangement INTEGRATION.md describes.  Run in a child process to keep sys.path clean."""
    code = r'''
import sys, types, torch
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/g_nerf"); sys.path.insert(0, %r)
tvr = types.ModuleType("torchvision.models.resnet"); tvr.ResNet = type("ResNet", (torch.nn.Module,), {}); tvr.Bottleneck = type("B", (torch.nn.Module,), {})
sys.modules.update({"torchvision": types.ModuleType("torchvision"), "torchvision.models": types.ModuleType("torchvision.models"), "torchvision.models.resnet": tvr})
import torch_utils.ops.bias_act as ba, torch_utils.persistence as pe, torch_utils.ops.conv2d_resample as cr, torch_utils.ops.conv2d_gradfix as gf, torch_utils.ops.fma as fm
import training.volumetric_rendering.renderer as rr, training.triplane as tp
assert "g-nerf_amd" in ba.__file__ and "g-nerf_amd" in rr.__file__, (ba.__file__, rr.__file__)
assert "/root/reference" in pe.__file__ and "/root/reference" in gf.__file__ and "/root/reference" in tp.__file__
assert "g-nerf_amd" in cr.__file__ and "g-nerf_amd" in fm.__file__ and cr.conv2d_gradfix is gf          # round 3: the two caller modules are overlaid too
assert tp.ImportanceRenderer is rr.ImportanceRenderer
import torch_utils.ops.upfirdn2d as up
assert cr.upfirdn2d is up and "g-nerf_amd" in up.__file__
dec = tp.OSGDecoder(32, {"decoder_lr_mul": 1, "decoder_output_dim": 32})
assert rr._osg_decoder_weights(dec) is not None
print("overlay ok")
''' % os.path.join(ROOT, 'g-nerf_amd')
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=env, cwd='/tmp')
    assert out.returncode == 0 and 'overlay ok' in out.stdout, out.stderr[-2000:]


def test_generator_host_side_caches_and_backend_switch():
    """Small host-side pieces of the generator harness: the per-parameter-version cache of converted parameters (the fp16 copies of
    fp32 biases), the orbit's last short block keeping the batch size, and the MIOpen solver-search switch."""
    import gnerf_generator as GG
    import gnerf_harness as H
    import gen_videos_mi355x as GV
    m = torch.nn.Module()
    m.bias = torch.nn.Parameter(torch.randn(8), requires_grad=False)
    a = GG._cast_param(m, 'bias', torch.float16)
    assert a.dtype == torch.float16 and GG._cast_param(m, 'bias', torch.float16) is a            # cached
    assert GG._cast_param(m, 'bias', torch.float32) is m.bias                                    # nothing to convert
    with torch.no_grad():
        m.bias.add_(1.0)
    b = GG._cast_param(m, 'bias', torch.float16)
    assert b is not a and torch.equal(b, m.bias.half())                                          # the version moved: converted again
    m.bias.requires_grad_(True)
    assert GG._cast_param(m, 'bias', torch.float16).requires_grad                                # training: stays in the graph, not cached
    m.none = None
    assert GG._cast_param(m, 'none', torch.float16) is None
    # orbit of 6 frames, 4 cameras per call: the second block is padded to 4 cameras and its 2 extra frames dropped
    calls = []

    class G:
        rendering_kwargs = {'avg_camera_radius': 2.7, 'depth_resolution': 4, 'depth_resolution_importance': 4}
        z_dim = 8

        def mapping(self, z, c):
            return torch.zeros(z.shape[0], 14, 8)

        def synthesis(self, ws, c, **kw):
            calls.append(c.shape[0])
            n = c.shape[0]
            return {'image': c[:, :3].reshape(n, 3, 1, 1).expand(n, 3, 4, 4), 'image_raw': torch.zeros(n, 3, 2, 2)}
    frames, raws, (lo, hi) = GV.render_orbit(G(), torch.zeros(1, 8), 6, 2, torch.device('cpu'), double_depth=False, frames_per_call=4)
    assert calls == [4, 4] and frames.shape == (6, 4, 4, 3) and raws.shape == (6, 2, 2, 3) and (lo, hi) == (0, 6)
    singles, _, _ = GV.render_orbit(G(), torch.zeros(1, 8), 6, 2, torch.device('cpu'), double_depth=False)
    assert torch.equal(frames, singles)
    # the orbit's camera labels in one batch == frame by frame, bit for bit
    for fn, frames in ((240, range(240)), (120, range(30, 60)), (7, [6, 0, 3])):
        want = torch.cat([H.camera_label(H.orbit_pose(i, fn, 2.7)) for i in frames])
        assert torch.equal(H.orbit_labels(frames, fn, 2.7), want)
    # solver search: argument wins over the environment, environment over the default (on)
    old = torch.backends.cudnn.benchmark
    try:
        assert H.configure_backend(False) is False and torch.backends.cudnn.benchmark is False
        assert H.configure_backend(True) is True and torch.backends.cudnn.benchmark is True
        os.environ['GNERF_MIOPEN_FIND'] = '0'
        assert H.configure_backend() is False
        del os.environ['GNERF_MIOPEN_FIND']
        assert H.configure_backend() is True
    finally:
        os.environ.pop('GNERF_MIOPEN_FIND', None)
        torch.backends.cudnn.benchmark = old


def test_c_abi_from_plain_c(tmp_path):
    """include/gnerf_hip.h is a C header (the boundary a cgo / JNI / ctypes binding would bind): tests/cabi/cabi_smoke.c compiles against
    it as strict C99 with gcc, links libgnerf_hip.so, and the entry points that need no GPU answer -- version = the header's, build
    string, workspace size, argument errors with a message.  The same struct size as the ctypes mirror."""
    import shutil
    import subprocess
    import gnerf_hip
    if shutil.which('gcc') is None or not os.path.isfile(gnerf_hip.LIB_PATH):
        pytest.skip('needs gcc and the built library')
    exe = str(tmp_path / 'cabi_smoke')
    libdir = os.path.dirname(gnerf_hip.LIB_PATH)
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-I' + os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'cabi', 'cabi_smoke.c'), '-o', exe, '-L' + libdir, '-lgnerf_hip',
                    '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'], check=True, capture_output=True, text=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f'abi={gnerf_hip.ABI_VERSION} ' in out.stdout
    assert f'sizeof(gnerf_render_params)={ctypes.sizeof(gnerf_hip.RenderParams)} ' in out.stdout


def test_bench_accounting_matches_survey():
    """bench.py's algorithmic-work constants are SURVEY.md section 8d's: 136 331 908 compulsory HBM bytes per config-2 call,
    8 320 MLP FLOP per sample, 1 536 gather bytes per sample; the scene builder gives config 2's shapes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.hbm_bytes_per_call(4, 4 * 128 * 128, 48, 48, 256) == 136331908
    assert bench.FLOP_MLP_PER_SAMPLE == 8320 and bench.GATHER_BYTES_PER_SAMPLE == 1536
    assert (bench.N_ITEMS, bench.RES, bench.S_COARSE, bench.S_FINE, bench.PLANE) == (4, 128, 48, 48, 256)
    planes, dec, c2w, intr = bench._scene(torch.device('cpu'), 0, n_items=2, plane=8)
    assert planes.shape == (2, 3, 32, 8, 8) and [tuple(t.shape) for t in dec] == [(64, 32), (64,), (33, 64), (33,)]
    assert c2w.shape == (2, 4, 4) and abs(float(c2w[0, :3, 3].norm()) - 2.7) < 1e-5 and float(intr[0, 0, 0]) == pytest.approx(4.2647)


def test_profiler_ranges_on_the_ops():
    """the reference wraps conv2d_resample (conv2d_resample.py:47) and its ops' reference paths in
    misc.profiled_function = record_function ranges.  The overlay keeps the range name; every native-op wrapper of gnerf_hip opens
    `gnerf_hip::<op>` -- only while a profiler is collecting (no host cost otherwise)."""
    import torch
    import gnerf_hip
    from torch_utils.ops import conv2d_resample
    x, w = torch.randn(1, 4, 8, 8), torch.randn(6, 4, 3, 3)
    assert not torch.autograd._profiler_enabled()
    want = conv2d_resample.conv2d_resample(x, w, padding=1)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        got = conv2d_resample.conv2d_resample(x, w, padding=1)
    assert torch.equal(got, want)
    assert 'conv2d_resample' in {e.name for e in prof.events()}
    # the native wrappers carry the decorator (their bodies need a GPU: checked by name and wrapping only)
    for op in ('bias_act', 'upfirdn2d', 'filtered_lrelu', 'render_forward', 'render_backward', 'query_points', 'modconv_epilogue', 'torgb_channels_last'):
        assert hasattr(getattr(gnerf_hip, op), '__wrapped__'), op
