"""Drop-in API check: every public callable of the overlay modules has the reference's signature (names, order, defaults), and
every public name the reference modules define exists in the overlay.  Needs the reference tree (build container only)."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/g_nerf'

SCRIPT = r'''
import sys, inspect, importlib, importlib.util, json
sys.dont_write_bytecode = True
MODS = ["torch_utils.ops.bias_act", "torch_utils.ops.upfirdn2d", "torch_utils.ops.filtered_lrelu", "torch_utils.ops.grid_sample_gradfix",
        "torch_utils.ops.conv2d_resample", "torch_utils.ops.fma", "torch_utils.custom_ops", "training.volumetric_rendering.renderer", "training.volumetric_rendering.ray_marcher",
        "training.volumetric_rendering.ray_sampler", "training.volumetric_rendering.math_utils"]


def load_from(root, name):
    path = root + "/" + name.replace(".", "/") + ".py"
    spec = importlib.util.spec_from_file_location("probe_" + name.replace(".", "_") + ("_ref" if "reference" in root else "_ours"), path,
                                                  submodule_search_locations=None)
    m = importlib.util.module_from_spec(spec)
    m.__package__ = name.rsplit(".", 1)[0]
    spec.loader.exec_module(m)
    return m


def public(m):
    out = {}
    for k, v in vars(m).items():
        if k.startswith("_") or inspect.ismodule(v):
            continue
        if getattr(v, "__module__", None) not in (m.__name__, None) and not isinstance(v, (int, float, str, bool, dict)):
            continue                                   # imported from elsewhere
        out[k] = v
    return out


def sig(f):
    try:
        s = inspect.signature(f)
    except (TypeError, ValueError):
        return None
    return [(p.name, str(p.kind), repr(p.default) if p.default is not inspect._empty else None) for p in s.parameters.values()]


report = {"missing": [], "signature": []}
sys.path.insert(0, %(ref)r)                            # the reference alone first ...
ref_mods = {n: importlib.import_module(n) for n in MODS}
ref_pub = {n: {k: (sig(v) if callable(v) else "value", {mn: sig(mv) for mn, mv in vars(v).items() if callable(mv) and not mn.startswith("_")} if inspect.isclass(v) else None)
               for k, v in public(m).items()} for n, m in ref_mods.items()}
for n in list(sys.modules):
    if n.split(".")[0] in ("torch_utils", "training", "dnnlib"):
        del sys.modules[n]
sys.path.insert(0, %(ours)r)                           # ... then the overlay in front of it
for n in MODS:
    m = importlib.import_module(n)
    assert %(ours)r in m.__file__, (n, m.__file__)
    ours = public(m)
    for k, (s_ref, methods_ref) in ref_pub[n].items():
        if k not in vars(m):
            report["missing"].append(n + "." + k)
            continue
        v = vars(m)[k]
        if s_ref not in (None, "value") and callable(v) and not inspect.isclass(v):
            if sig(v) != s_ref:
                report["signature"].append([n + "." + k, sig(v), s_ref])
        if methods_ref:
            for mn, ms in methods_ref.items():
                mv = getattr(v, mn, None)
                if mv is None:
                    report["missing"].append(n + "." + k + "." + mn)
                elif ms is not None and sig(mv) != ms:
                    report["signature"].append([n + "." + k + "." + mn, sig(mv), ms])
print(json.dumps(report))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_overlay_matches_reference_signatures():
    import json
    code = SCRIPT % dict(ref=REF, ours=os.path.join(ROOT, 'g-nerf_amd'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), cwd='/tmp', timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    report = json.loads(r.stdout.strip().splitlines()[-1])
    assert report['missing'] == [], report['missing']
    assert report['signature'] == [], report['signature']


ERR_SCRIPT = r'''
import sys, json
sys.dont_write_bytecode = True
for p in reversed(%(paths)r): sys.path.insert(0, p)
import torch, numpy as np
from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu
x = torch.randn(2, 3, 8, 8)
f = upfirdn2d.setup_filter([1, 3, 3, 1])
cases = {
 'ba_bad_impl': lambda: bias_act.bias_act(x, impl='foo'), 'ba_neg_clamp': lambda: bias_act.bias_act(x, clamp=-1),
 'ba_bad_b_shape': lambda: bias_act.bias_act(x, torch.zeros(4)), 'ba_bad_b_rank': lambda: bias_act.bias_act(x, torch.zeros(3, 1)),
 'ba_bad_dim': lambda: bias_act.bias_act(x, torch.zeros(3), dim=7), 'ba_bad_act': lambda: bias_act.bias_act(x, act='nope'),
 'up_bad_impl': lambda: upfirdn2d.upfirdn2d(x, f, impl='foo'), 'up_bad_rank': lambda: upfirdn2d.upfirdn2d(x[0], f),
 'up_bad_filter_dtype': lambda: upfirdn2d.upfirdn2d(x, f.double()), 'up_bad_filter_rank': lambda: upfirdn2d.upfirdn2d(x, f[None]),
 'up_bad_pad_len': lambda: upfirdn2d.upfirdn2d(x, f, padding=[1, 2, 3]), 'up_bad_pad_type': lambda: upfirdn2d.upfirdn2d(x, f, padding=[1.0, 2]),
 'up_bad_up': lambda: upfirdn2d.upfirdn2d(x, f, up=0), 'up_float_up': lambda: upfirdn2d.upfirdn2d(x, f, up=1.5), 'up_bad_up_len': lambda: upfirdn2d.upfirdn2d(x, f, up=[1, 2, 3]),
 'setup_filter_3d': lambda: upfirdn2d.setup_filter(np.ones((2, 2, 2))), 'setup_filter_empty': lambda: upfirdn2d.setup_filter([]),
 'setup_filter_none': lambda: tuple(upfirdn2d.setup_filter(None).shape), 'setup_filter_sep8': lambda: tuple(upfirdn2d.setup_filter(list(range(1, 9))).shape),
 'filter2d': lambda: tuple(upfirdn2d.filter2d(x, f).shape), 'upsample2d': lambda: tuple(upfirdn2d.upsample2d(x, f, up=2).shape),
 'downsample2d': lambda: tuple(upfirdn2d.downsample2d(x, f, down=2).shape), 'upsample2d_xy': lambda: tuple(upfirdn2d.upsample2d(x, f, up=[2, 1], padding=[1, 0]).shape),
 'fl_bad_impl': lambda: filtered_lrelu.filtered_lrelu(x, impl='foo'), 'fl_bad_up': lambda: filtered_lrelu.filtered_lrelu(x, up=0),
 'fl_bad_gain': lambda: filtered_lrelu.filtered_lrelu(x, gain=-1), 'fl_bad_b': lambda: filtered_lrelu.filtered_lrelu(x, b=torch.zeros(5)),
 'fl_bad_clamp': lambda: filtered_lrelu.filtered_lrelu(x, clamp=-2), 'fl_bad_pad_len': lambda: filtered_lrelu.filtered_lrelu(x, padding=[1, 2, 3]),
 'fl_plain': lambda: tuple(filtered_lrelu.filtered_lrelu(x).shape),
}
out = {}
for k, fn in cases.items():
    try:
        out[k] = 'ok:' + str(fn())
    except Exception as e:
        out[k] = type(e).__name__
print(json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_overlay_error_behaviour_matches_reference():
    """The same misuse raises the same exception type in the overlay ops as in the reference's (and valid helper calls give the same
    shapes): the error behaviour is part of the drop-in contract."""
    import json
    res = {}
    for name, paths in (('ref', [REF]), ('ours', [os.path.join(ROOT, 'g-nerf_amd'), REF])):
        r = subprocess.run([sys.executable, '-c', ERR_SCRIPT % dict(paths=paths)], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), cwd='/tmp', timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['ours'] == res['ref'], {k: (res['ref'][k], res['ours'][k]) for k in res['ref'] if res['ref'][k] != res['ours'][k]}


RENDER_SCRIPT = r'''
import sys, json
sys.dont_write_bytecode = True
for p in reversed(%(paths)r): sys.path.insert(0, p)
import numpy as np, torch
from training.volumetric_rendering.renderer import ImportanceRenderer, sample_from_3dgrid, generate_planes, project_onto_planes
from training.volumetric_rendering.ray_sampler import RaySampler
from training.volumetric_rendering.ray_marcher import MipRayMarcher2
from training.volumetric_rendering import math_utils


class FC(torch.nn.Module):
    def __init__(self, i, o, g):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(o, i, generator=g)); self.bias = torch.nn.Parameter(torch.randn(o, generator=g) * 0.1)
        self.activation = 'linear'; self.weight_gain = 1 / np.sqrt(i); self.bias_gain = 1
    def forward(self, x):
        return torch.addmm((self.bias * self.bias_gain).unsqueeze(0), x, (self.weight * self.weight_gain).t())


class Dec(torch.nn.Module):
    def __init__(self, g):
        super().__init__()
        self.net = torch.nn.Sequential(FC(32, 64, g), torch.nn.Softplus(), FC(64, 33, g))
    def forward(self, feats, dirs):
        x = feats.mean(1); N, M, C = x.shape
        x = self.net(x.view(N * M, C)).view(N, M, -1)
        return {'rgb': torch.sigmoid(x[..., 1:]) * 1.002 - 0.001, 'sigma': x[..., 0:1]}


g = torch.Generator().manual_seed(0)
planes = torch.randn(2, 3, 32, 12, 12, generator=g)
dec = Dec(g)
cams = torch.eye(4).repeat(2, 1, 1); cams[:, 2, 3] = 2.7; cams[1, 0, 3] = 0.3; cams[:, 2, 2] = -1; cams[:, 0, 0] = -1
intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(2, 1, 1)
o, d = RaySampler()(cams, intr, 6)
base = dict(depth_resolution=10, depth_resolution_importance=9, ray_start=2.25, ray_end=3.3, box_warp=1, clamp_mode='softplus', disparity_space_sampling=False)
out = {'rays_o': o.numpy().tolist(), 'rays_d': d.numpy().tolist()}
variants = {'base': {}, 'white_back': {'white_back': True}, 'disparity': {'disparity_space_sampling': True}, 'auto': {'ray_start': 'auto', 'ray_end': 'auto'},
            'no_importance': {'depth_resolution_importance': 0}, 'noise': {'density_noise': 0.3}, 'box2': {'box_warp': 2}}
ren = ImportanceRenderer()
for name, kw in variants.items():
    torch.manual_seed(5)
    with torch.no_grad():
        r = ren(planes, dec, o, d, dict(base, **kw))
    out[name] = [t.numpy().tolist() for t in r]
torch.manual_seed(6)
pts = torch.rand(2, 50, 3, generator=g) * 1.2 - 0.6
with torch.no_grad():
    rm = ren.run_model(planes, dec, pts, torch.zeros_like(pts), base)
out['run_model'] = [rm['rgb'].numpy().tolist(), rm['sigma'].numpy().tolist()]
out['grid3d'] = sample_from_3dgrid(torch.randn(1, 4, 5, 6, 7, generator=g), torch.rand(2, 9, 3, generator=g) * 2 - 1).numpy().tolist()
out['project'] = project_onto_planes(generate_planes(), pts).numpy().tolist()
v = torch.randn(4, 7, 3, generator=g)
out['normalize'] = math_utils.normalize_vecs(v).numpy().tolist()
out['dot'] = math_utils.torch_dot(v, v.flip(0)).numpy().tolist()
lo, hi = math_utils.get_ray_limits_box(o, d, box_side_length=1)
out['limits'] = [torch.nan_to_num(lo, posinf=9, neginf=-9).numpy().tolist(), torch.nan_to_num(hi, posinf=9, neginf=-9).numpy().tolist()]
out['linspace'] = math_utils.linspace(torch.tensor([0.0, 1.0]), torch.tensor([2.0, 5.0]), 7).numpy().tolist()
m = MipRayMarcher2()
cols, dens, deps = torch.rand(2, 5, 8, 3, generator=g), torch.randn(2, 5, 8, 1, generator=g), torch.sort(torch.rand(2, 5, 8, 1, generator=g) + 2, dim=2)[0]
out['marcher'] = [t.numpy().tolist() for t in m(cols, dens, deps, base)]
try:
    m(cols, dens, deps, dict(base, clamp_mode='relu')); out['marcher_bad_mode'] = 'ok'
except Exception as e:
    out['marcher_bad_mode'] = type(e).__name__
print(json.dumps(out))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_overlay_renderer_modules_match_reference_on_cpu():
    """The four renderer modules of the overlay against the reference's on the CPU, same seeds (both consume torch's generator in the
    same order): ImportanceRenderer.forward under every option the reference reads (white_back, disparity sampling, 'auto' ray limits,
    no importance pass, density noise, box_warp), run_model, sample_from_3dgrid, project_onto_planes, the math_utils helpers and
    MipRayMarcher2 incl. its refusal of other clamp modes."""
    import json
    import numpy as np
    res = {}
    for name, paths in (('ref', [REF]), ('ours', [os.path.join(ROOT, 'g-nerf_amd'), REF])):
        r = subprocess.run([sys.executable, '-c', RENDER_SCRIPT % dict(paths=paths)], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), cwd='/tmp', timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['ours'].keys() == res['ref'].keys()
    assert res['ours']['marcher_bad_mode'] == res['ref']['marcher_bad_mode']
    for k in res['ref']:
        if k == 'marcher_bad_mode':
            continue
        a, b = res['ref'][k], res['ours'][k]
        if not isinstance(a[0], list) or not isinstance(a[0][0], list) or isinstance(a[0][0][0], float):
            a, b = [a], [b]
        for x, y in zip(a, b):
            np.testing.assert_allclose(np.asarray(y, dtype=np.float64), np.asarray(x, dtype=np.float64), rtol=2e-4, atol=2e-5, err_msg=k)


OPS_SCRIPT = r'''
import sys, json
sys.dont_write_bytecode = True
for p in reversed(%(paths)r): sys.path.insert(0, p)
import numpy as np, torch
from torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu, grid_sample_gradfix
rng = np.random.default_rng(3)
out = {}
ACTS = list(bias_act.activation_funcs.keys())
out['acts'] = ACTS
out['act_table'] = {k: [v.def_alpha, v.def_gain, v.cuda_idx, v.ref, v.has_2nd_grad] for k, v in bias_act.activation_funcs.items()}
for i, act in enumerate(ACTS):
    x = torch.from_numpy(rng.standard_normal((2, 5, 4, 3))).double().requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal(5)).double().requires_grad_(True)
    kw = dict(dim=1, act=act, alpha=(None if i %% 2 else 0.3), gain=(None if i %% 3 else 1.7), clamp=(None if i %% 2 == 0 else 0.9))
    y = bias_act.bias_act(x, b, **kw)
    w = torch.from_numpy(rng.standard_normal(tuple(y.shape))).double()
    gx, gb = torch.autograd.grad((y * w).sum(), (x, b), create_graph=True)
    s2 = (gx * w).sum() + (gb ** 2).sum()
    ggx = torch.autograd.grad(s2, x, allow_unused=True)[0] if s2.requires_grad else None
    out['ba_' + act] = [y.detach().numpy().tolist(), gx.detach().numpy().tolist(), gb.detach().numpy().tolist(), (torch.zeros_like(x) if ggx is None else ggx).numpy().tolist()]
for j, (up, down, pad, ftaps, flip, gain) in enumerate([(1, 1, [1, 1, 1, 1], [1, 3, 3, 1], False, 4.0), (2, 1, [2, 1, 2, 1], [1, 3, 3, 1], False, 4.0), (1, 2, [1, 1, 1, 1], [1, 3, 3, 1], True, 1.0),
                                                      ([2, 1], [1, 3], [0, 1, 2, 0], list(range(1, 9)), False, 2.0), (3, 2, 2, [1, 2, 1], True, 1.5), (1, 1, [-1, 0, 0, -1], None, False, 1.0)]):
    x = torch.from_numpy(rng.standard_normal((2, 3, 9, 11))).float().requires_grad_(True)
    f = upfirdn2d.setup_filter(ftaps) if ftaps is not None else None
    y = upfirdn2d.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    w = torch.from_numpy(rng.standard_normal(tuple(y.shape))).float()
    gx, = torch.autograd.grad((y * w).sum(), x)
    out['up_%%d' %% j] = [y.detach().numpy().tolist(), gx.numpy().tolist()]
for j, kw in enumerate([dict(), dict(up=2, down=2, padding=[5, 5, 5, 5], gain=1.3, slope=0.1, clamp=0.8), dict(up=2, down=1, padding=[3, 2, 3, 2], flip_filter=True), dict(up=1, down=2, padding=3)]):
    x = torch.from_numpy(rng.standard_normal((2, 3, 8, 8))).float().requires_grad_(True)
    b = torch.from_numpy(rng.standard_normal(3)).float().requires_grad_(True)
    fu = upfirdn2d.setup_filter([1, 4, 6, 4, 1, 0][: 6]) if kw.get('up', 1) > 1 else None
    fd = upfirdn2d.setup_filter([1, 3, 3, 1, 2, 2]) if kw.get('down', 1) > 1 else None
    y = filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw)
    w = torch.from_numpy(rng.standard_normal(tuple(y.shape))).float()
    gx, gb = torch.autograd.grad((y * w).sum(), (x, b))
    out['fl_%%d' %% j] = [y.detach().numpy().tolist(), gx.numpy().tolist(), gb.numpy().tolist()]
img = torch.from_numpy(rng.standard_normal((2, 3, 6, 7))).float().requires_grad_(True)
grid = torch.from_numpy(rng.uniform(-1.2, 1.2, (2, 4, 5, 2))).float().requires_grad_(True)
w = torch.from_numpy(rng.standard_normal((2, 3, 4, 5))).float()
for flag in (False, True):
    grid_sample_gradfix.enabled = flag
    try:            # (with the flag on, the reference's own op fails under torch 2.x: torch._C._jit_get_operation returns a tuple there)
        y = grid_sample_gradfix.grid_sample(img, grid)
        gi, gg = torch.autograd.grad((y * w).sum(), (img, grid))
        out['gs_%%d' %% flag] = [y.detach().numpy().tolist(), gi.numpy().tolist(), gg.numpy().tolist()]
    except TypeError:
        out['gs_%%d' %% flag] = None
grid_sample_gradfix.enabled = False
print(json.dumps(out))
'''


def _close(a, b, key, rtol=2e-5, atol=2e-6):
    import numpy as np
    if isinstance(a, list) and a and isinstance(a[0], list) and not _is_numeric(a):
        assert len(a) == len(b), key
        for x, y in zip(a, b):
            _close(x, y, key, rtol, atol)
    else:
        np.testing.assert_allclose(np.asarray(b, dtype=np.float64), np.asarray(a, dtype=np.float64), rtol=rtol, atol=atol, err_msg=key)


def _is_numeric(a):
    import numpy as np
    try:
        return np.asarray(a, dtype=np.float64).dtype == np.float64
    except (ValueError, TypeError):
        return False


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_overlay_ops_match_reference_on_cpu():
    """The four op modules of the overlay against the reference's on the CPU (the reference's `_ref` forms; first- and second-order
    gradients through autograd): every activation of bias_act incl. the activation table itself, upfirdn2d variants (per-axis factors,
    crops, separable 8-tap filter, no filter), filtered_lrelu, grid_sample with the flag off and on."""
    import json
    res = {}
    for name, paths in (('ref', [REF]), ('ours', [os.path.join(ROOT, 'g-nerf_amd'), REF])):
        r = subprocess.run([sys.executable, '-c', OPS_SCRIPT % dict(paths=paths)], capture_output=True, text=True,
                           env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), cwd='/tmp', timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[name] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res['ours']['acts'] == res['ref']['acts'] and res['ours']['act_table'] == res['ref']['act_table']
    assert res['ours']['gs_1'] is not None
    if res['ref']['gs_1'] is None:                   # the reference's flag-on path is broken under this torch: the flag-off results are the same function
        res['ref']['gs_1'] = res['ref']['gs_0']
    for k in res['ref']:
        if k in ('acts', 'act_table'):
            continue
        for x, y in zip(res['ref'][k], res['ours'][k]):
            _close(x, y, k)
