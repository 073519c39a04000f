"""The REFERENCE's TriPlaneGenerator run with this repo's overlay in front of it on sys.path -- torch_utils.ops.{conv2d_resample, fma,
bias_act, upfirdn2d}, torch_utils.custom_ops and training.volumetric_rendering.* resolve to g-nerf_amd/, every other module
(training.triplane, networks_stylegan2, superresolution, ...) to /root/reference/g_nerf -- against the fixture that
tests/golden/make_golden.py made from the reference alone (generator_n4.npz, BASELINE config 3 at N=4).  This is what a G-NeRF
checkout runs after the swap INTEGRATION.md describes.  Needs the reference tree: build container only, CPU, fp32."""

import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/g_nerf'

SCRIPT = r'''
import os, sys, types, numpy as np, torch
sys.dont_write_bytecode = True
for p in reversed(%(paths)r): sys.path.insert(0, p)
tvr = types.ModuleType("torchvision.models.resnet"); tvr.ResNet = type("ResNet", (torch.nn.Module,), {}); tvr.Bottleneck = type("B", (torch.nn.Module,), {})
sys.modules.update({"torchvision": types.ModuleType("torchvision"), "torchvision.models": types.ModuleType("torchvision.models"), "torchvision.models.resnet": tvr})
sys.path.insert(0, os.path.join(%(root)r, "tests", "golden"))
import det_init as DI, make_golden as MG
import torch_utils.ops.conv2d_resample as CR, torch_utils.ops.fma as FMA, torch_utils.ops.bias_act as BA, torch_utils.ops.upfirdn2d as UF
import torch_utils.ops.conv2d_gradfix as GF, training.triplane as TP, training.networks_stylegan2 as NS, training.volumetric_rendering.renderer as RR
ours, ref = os.path.join(%(root)r, "g-nerf_amd"), "/root/reference"
for m in (CR, FMA, BA, UF, RR): assert m.__file__.startswith(ours), m.__file__
for m in (GF, TP, NS): assert m.__file__.startswith(ref), m.__file__
assert NS.conv2d_resample is CR and NS.fma is FMA and CR.conv2d_gradfix is GF        # the reference's layers call the overlay's modules
calls = {"conv": 0, "fma": 0}
_c, _f = CR.conv2d_resample, FMA.fma
def conv(*a, **k): calls["conv"] += 1; return _c(*a, **k)
def fma(*a, **k): calls["fma"] += 1; return _f(*a, **k)
CR.conv2d_resample, FMA.fma = conv, fma
torch.set_num_threads(os.cpu_count() or 1)
torch.manual_seed(0)
G = DI.det_init_(TP.TriPlaneGenerator(**MG._ffhq_g_kwargs()), "G/").eval().requires_grad_(False)
batch = DI.synthetic_batch(4)
with torch.no_grad(), DI.DetNoise("config3"):
    ws = G.mapping(batch["z"], batch["c"])
    out = G.synthesis(ws, batch["c"], noise_mode="const", neural_rendering_resolution=64)
n_inference = dict(calls)
# the un-fused form (training mode: fused_modconv_default = 'inference_only') goes through fma as well
G.train()
with torch.no_grad(), DI.DetNoise("config5"):
    out_t = G.synthesis(ws[:1], batch["c"][:1], neural_rendering_resolution=64)
img = out["image"]
np.savez(%(out)r, ws_first=ws[:, 0, :8].numpy(), image_raw=out["image_raw"].numpy(), image_depth=out["image_depth"].numpy(),
         image_sub=img[:, :, 4::8, 4::8].numpy(), image_mean=img.mean((1, 2, 3)).numpy(), conv_calls=n_inference["conv"], fma_calls=calls["fma"],
         train_finite=bool(torch.isfinite(out_t["image"]).all()))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree only exists in the build container')
def test_reference_generator_through_overlay_matches_fixture(tmp_path, golden):
    out = str(tmp_path / 'overlay_ref.npz')
    code = SCRIPT % dict(paths=[os.path.join(ROOT, 'g-nerf_amd'), REF], root=ROOT, out=out)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'), cwd='/tmp', timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d, g = np.load(out), golden('generator_n4.npz')
    # 13 backbone + 6 superresolution modulated convolutions and 7 + 3 ToRGB layers reach conv2d_resample; fma only in training mode
    assert int(d['conv_calls']) == 29 and int(d['fma_calls']) > 0 and bool(d['train_finite'])
    np.testing.assert_allclose(d['ws_first'], g['ws_first'], atol=1e-6)
    for k in ('image_raw', 'image_depth', 'image_sub'):
        assert float(((d[k] - g[k]) ** 2).mean()) < 1e-10, k           # CPU fp32 on both sides: the same PyTorch ops in the same order
    np.testing.assert_allclose(d['image_mean'], g['image_mean'], atol=1e-5)
