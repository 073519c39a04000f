"""gnerf_generator.Generator (the inference-only stand-in for the hot path's callers, used where the reference tree is
absent) against the reference's own TriPlaneGenerator: same parameter / buffer names (strict state_dict load) and the same
images for the same weights, latent, camera and seed.  Needs the reference tree: build container only, CPU, fp32."""

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/g_nerf'

SCRIPT = r'''
import sys, types, numpy as np, torch
sys.dont_write_bytecode = True
for p in reversed(%(paths)r): sys.path.insert(0, p)
tvr = types.ModuleType("torchvision.models.resnet"); tvr.ResNet = type("ResNet", (torch.nn.Module,), {}); tvr.Bottleneck = type("B", (torch.nn.Module,), {})
sys.modules.update({"torchvision": types.ModuleType("torchvision"), "torchvision.models": types.ModuleType("torchvision.models"), "torchvision.models.resnet": tvr})
import gen_videos_mi355x as gv, gnerf_generator, gnerf_harness as H
import training.triplane
assert "/root/reference" in training.triplane.__file__
torch.set_num_threads(8)
G_ref = gv.build_random_generator(0, torch.device("cpu"))
assert type(G_ref).__name__ == "TriPlaneGenerator" and not isinstance(G_ref, gnerf_generator.Generator)
# make the comparison bite: non-zero noise strengths and biases (both are zero at initialisation)
g = torch.Generator().manual_seed(5)
with torch.no_grad():
    for n, p_ in G_ref.named_parameters():
        if n.endswith("noise_strength") or n.endswith(".bias"):
            p_.add_(torch.randn(p_.shape, generator=g) * 0.1)
G = gnerf_generator.Generator(rendering_kwargs=G_ref.rendering_kwargs).eval().requires_grad_(False)
missing = G.load_state_dict(G_ref.state_dict(), strict=True)
z = torch.randn(2, 512, generator=g)
c = torch.cat([H.camera_label(H.orbit_pose(i, 120)) for i in (3, 40)])
with torch.no_grad():
    ws_ref, ws = G_ref.mapping(z, c), G.mapping(z, c)
    torch.manual_seed(11); a = G_ref.synthesis(ws_ref, c, noise_mode="const", neural_rendering_resolution=64)
    torch.manual_seed(11); b = G.synthesis(ws, c, noise_mode="const", neural_rendering_resolution=64)
    planes_ref = G_ref.backbone.synthesis(ws_ref, noise_mode="const"); planes = G.backbone.synthesis(ws, noise_mode="const")
np.savez(%(out)r, ws=(ws_ref - ws).abs().max().numpy(), ws_shape=np.array(ws.shape), planes=(planes_ref - planes).abs().max().numpy(), planes_scale=planes_ref.abs().max().numpy(),
         **{k + "_err": (a[k] - b[k]).abs().max().numpy() for k in a}, **{k + "_scale": a[k].abs().max().numpy() for k in a}, image_shape=np.array(b["image"].shape))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_generator_matches_reference(tmp_path):
    out = str(tmp_path / 'cmp.npz')
    code = SCRIPT % dict(paths=[os.path.join(ROOT, 'g-nerf_amd'), REF], out=out)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), cwd='/tmp', timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    assert tuple(d['ws_shape']) == (2, 14, 512) and tuple(d['image_shape']) == (2, 3, 512, 512)
    assert float(d['ws']) < 1e-5
    assert float(d['planes']) < 1e-4 * max(1.0, float(d['planes_scale']))
    for k in ('image', 'image_raw', 'image_depth'):
        assert float(d[k + '_err']) < 2e-4 * max(1.0, float(d[k + '_scale'])), (k, float(d[k + '_err']), float(d[k + '_scale']))


SHAPES_SCRIPT = r'''
import sys, types, numpy as np, torch
sys.dont_write_bytecode = True
for p in reversed(%(paths)r): sys.path.insert(0, p)
tvr = types.ModuleType("torchvision.models.resnet"); tvr.ResNet = type("ResNet", (torch.nn.Module,), {}); tvr.Bottleneck = type("B", (torch.nn.Module,), {})
stubs = {"torchvision": types.ModuleType("torchvision"), "torchvision.models": types.ModuleType("torchvision.models"), "torchvision.models.resnet": tvr}
for name in ("cv2", "imageio", "mrcfile", "scipy.interpolate"):        # the reference CLI's imports that this container lacks or does not need
    stubs.setdefault(name, types.ModuleType(name))
sys.modules.update(stubs)
import gen_videos as ref_cli                      # the reference's gen_videos.py, for its create_samples()
import gen_videos_mi355x as gv, gnerf_generator
res = {}
for n, cube in ((16, 1.0), (37, 2.0), (300, 1.0)):
    want = ref_cli.create_samples(N=n, voxel_origin=[0, 0, 0], cube_length=cube)[0]
    lo = 0 if n < 300 else n ** 3 - 4000                                # the large lattice: only its tail (indices above 2^24)
    got = gv.voxel_samples(lo, n ** 3, n, cube, torch.device("cpu"))
    res["samples_%%d" %% n] = float((got - want[:, lo:]).abs().max())
# the density volume: chunked run_model on cached planes == the reference-style loop over sample_mixed
torch.manual_seed(0)
G = gnerf_generator.Generator().eval().requires_grad_(False)
ws = G.mapping(torch.randn(1, 512), torch.zeros(1, 25))
n = 20
vol = gv.extract_density_grid(G, ws, resolution=n, max_batch=3000, crop=False)
samples = ref_cli.create_samples(N=n, voxel_origin=[0, 0, 0], cube_length=G.rendering_kwargs["box_warp"])[0]
dirs = torch.zeros_like(samples); dirs[..., -1] = -1
with torch.no_grad():
    sig = G.sample_mixed(samples, dirs, ws, noise_mode="const")["sigma"].reshape(n, n, n).flip(0)
res["volume"] = float((vol - sig).abs().max()); res["volume_scale"] = float(sig.abs().max())
c = gv.extract_density_grid(G, ws, resolution=n, max_batch=3000, crop=True)
res["crop_ok"] = float(c[:2].abs().max() + c[-2:].abs().max() + c[:, :2].abs().max() + c[:, -2:].abs().max() + c[:, :, :2].abs().max() + c[:, :, -2:].abs().max())
res["crop_inner"] = float((c[3:-3, 3:-3, 3:-3] - vol[3:-3, 3:-3, 3:-3]).abs().max())
np.savez(%(out)r, **res)
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_density_volume_matches_reference_lattice(tmp_path):
    """gen_videos.py --shapes counterpart: the lattice points equal the reference's create_samples() (float-division quirk and
    fp32 index rounding included) and the chunked extraction equals the per-chunk sample_mixed loop."""
    out = str(tmp_path / 'shapes.npz')
    code = SHAPES_SCRIPT % dict(paths=[os.path.join(ROOT, 'g-nerf_amd'), REF], out=out)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), cwd='/tmp', timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    for n in (16, 37, 300):
        assert float(d[f'samples_{n}']) == 0.0, (n, float(d[f'samples_{n}']))
    assert float(d['volume']) <= 1e-5 * max(1.0, float(d['volume_scale']))
    assert float(d['crop_ok']) == 0.0 and float(d['crop_inner']) == 0.0


def test_clear_latent_caches():
    """gnerf_generator.clear_latent_caches drops the per-latent / per-parameter caches the fast path hangs on the modules (ADVICE r3:
    several hundred MB over the backbone with no release hook)."""
    import torch
    import gnerf_generator as GG
    m = torch.nn.Linear(2, 2)
    m.__dict__['_gnerf_latent_cache'] = {'x': (None, None, torch.zeros(3))}
    m.__dict__['_gnerf_prenorm'] = {'k': (None, torch.zeros(3))}
    root = torch.nn.Sequential(m)
    assert len(GG.latent_cache_tensors(root)) == 1
    GG.clear_latent_caches(root)
    assert '_gnerf_latent_cache' not in m.__dict__ and '_gnerf_prenorm' not in m.__dict__ and GG.latent_cache_tensors(root) == []
