#!/usr/bin/env python3
"""Randomised parity sweep of gnerf_render_backward against autograd through the fp64 CPU oracle.
usage: python tests/parity_tools/fuzz_backward.py [n_cases] [seed]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), os.path.join(ROOT, 'tests'), ROOT]
import numpy as np
import torch
import gnerf_hip
from test_gpu_parity import _random_scene, _oracle_grads, _rel

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda', 0)
worst, fails = 0.0, []
for case in range(n_cases):
    S = int(rng.choice([rng.integers(4, 49), rng.integers(49, 97), rng.integers(97, 130)], p=[0.5, 0.4, 0.1]))
    F = int(rng.choice([0, rng.integers(1, 49), rng.integers(49, 97)], p=[0.15, 0.45, 0.4]))
    N, res = int(rng.integers(1, 4)), int(rng.integers(2, 6))
    hw = (int(rng.integers(4, 24)), int(rng.integers(4, 24)))
    white_back = bool(rng.integers(0, 2))
    planes, dec, o, d, nc, nf = _random_scene(int(rng.integers(1 << 30)), N, res, S, F, hw)
    M = res * res
    gen = torch.Generator().manual_seed(case)
    g_rgb, g_depth, g_w = torch.randn(N, M, 32, generator=gen), torch.randn(N, M, 1, generator=gen), torch.randn(N, M, 1, generator=gen)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus', white_back=white_back)
    ref_planes, ref_dec = _oracle_grads(planes, dec, o, d, nc, nf, opts, g_rgb, g_depth, g_w)
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    gp, gdec = gnerf_hip.render_backward(nhwc, N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None,
                                         g_rgb.to(dev), g_depth.to(dev), g_w.to(dev), depth_resolution=S, depth_resolution_importance=F,
                                         ray_start=2.25, ray_end=3.3, box_warp=1.0, white_back=white_back, image_width=res)
    errs = [_rel(gnerf_hip.planes_from_nhwc(gp, N).cpu(), ref_planes)] + [_rel(a.cpu(), b) for a, b in zip(gdec, ref_dec)]
    worst = max(worst, max(errs))
    if max(errs) > 2e-3:
        fails.append(dict(case=case, S=S, F=F, N=N, res=res, hw=hw, white_back=white_back, errs=errs))
print(json.dumps({'cases': n_cases, 'worst_rel_err': worst, 'failures': fails}))
sys.exit(1 if fails else 0)
