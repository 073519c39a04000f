#!/usr/bin/env python3
"""Randomised parity sweep of the fused renderer against the CPU oracle: random sample counts (all three kernels and the three
pipelined instantiations), ragged ray counts, plane sizes, white_back, disparity-space sampling and per-ray limits; a third of
the cases with plane / decoder magnitudes drawn log-uniformly from 1e-4..1e4 (the device-side choice between the f16 hi/lo and
the exact-fp32 decoder arithmetic; tolerance tied to the fp32 noise floor measured with the float64 oracle), half of the
cases with the planes in the interleaved [N,H,W,96] layout.
usage: python tests/parity_tools/fuzz_render.py [n_cases] [seed]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), os.path.join(ROOT, 'tests'), ROOT]
import numpy as np
import torch
import gnerf_hip
from oracle import render_ref as R
from test_gpu_parity import _random_scene

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda', 0)
worst = {'mse': 0.0, 'depth': 0.0, 'wsum': 0.0}
fails = []
only = set(int(v) for v in os.environ['FUZZ_ONLY'].split(',')) if os.environ.get('FUZZ_ONLY') else None     # re-run these cases of the sequence
for case in range(n_cases):
    S = int(rng.choice([rng.integers(4, 49), rng.integers(49, 97), rng.integers(97, 145), rng.integers(145, 200)], p=[0.35, 0.4, 0.2, 0.05]))
    F = int(rng.choice([0, rng.integers(1, 49), rng.integers(49, 97), rng.integers(97, 145), rng.integers(145, 180)], p=[0.1, 0.35, 0.35, 0.15, 0.05]))
    N, res = int(rng.integers(1, 4)), int(rng.integers(2, 7))
    hw = (int(rng.integers(4, 40)), int(rng.integers(4, 40)))
    white_back, disparity = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) == 0)
    per_ray = (not disparity) and bool(rng.integers(0, 4) == 0)
    planes, dec, o, d, nc, nf = _random_scene(int(rng.integers(1 << 30)), N, res, S, F, hw)
    wild = bool(rng.integers(0, 3) == 0)
    if wild:
        planes = planes * float(10 ** rng.uniform(-4, 4))
        ws = float(10 ** rng.uniform(-3, 3))
        dec = [t * ws for t in dec]
    interleaved = bool(rng.integers(0, 2))
    if only is not None and case not in only:
        continue
    rs, re = 2.25, 3.3
    if per_ray:
        g = torch.Generator().manual_seed(case)
        rs = 2.0 + 0.5 * torch.rand(N, res * res, 1, generator=g)
        re = rs + 0.6 + 0.6 * torch.rand(N, res * res, 1, generator=g)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=rs, ray_end=re, box_warp=1.0, clamp_mode='softplus',
                white_back=white_back, disparity_space_sampling=disparity)
    ref_rgb, ref_depth, ref_w = R.render(planes, dec, o, d, opts, nc, nf)
    floor = w_floor = 0.0
    if wild:
        dd = lambda t: t.double() if isinstance(t, torch.Tensor) else t
        ex_rgb, _, ex_w = R.render(planes.double(), [t.double() for t in dec], o.double(), d.double(), dict(opts, ray_start=dd(rs), ray_end=dd(re)), nc.double(), nf.double())
        floor, w_floor = float(((ref_rgb.double() - ex_rgb) ** 2).mean()), float((ref_w.double() - ex_w).abs().max())
    if interleaved:
        nhwc = planes.to(dev).reshape(N, 96, *hw).permute(0, 2, 3, 1).contiguous()
    else:
        nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    to = lambda t: t.to(dev) if isinstance(t, torch.Tensor) else t
    rgb, depth, wsum = gnerf_hip.render_forward(nhwc, N, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev) if F else None,
                                                depth_resolution=S, depth_resolution_importance=F, ray_start=to(rs), ray_end=to(re), box_warp=1.0,
                                                white_back=white_back, disparity_space_sampling=disparity, image_width=res)
    mse = float(((rgb.cpu() - ref_rgb) ** 2).mean())
    # depth = sum(w t) / sum(w): where a ray's weight sum is tiny (wild magnitudes: density ~ 0) the quotient is ill-conditioned in fp32 --
    # the fp32 oracle itself is then off by ~4e-7 in the weight sum against the float64 oracle -- so depth is compared on rays with
    # a weight sum of at least 1e-2 (seed 31 found six such rays in 1 500 cases: weight sums of 3e-7 .. 3e-4, depth off by 1e-3 .. 4e-2)
    well = ref_w >= 1e-2
    de = float(((depth.cpu() - ref_depth).abs() * well).max())
    we = float((wsum.cpu() - ref_w).abs().max())
    worst = {'mse': max(worst['mse'], mse), 'depth': max(worst['depth'], de), 'wsum': max(worst['wsum'], we)}
    if not torch.isfinite(rgb).all() or not torch.isfinite(wsum).all():
        mse = float('inf')
    if floor >= 1e-9:
        de = 0.0                                # depth of an ill-conditioned scene is not compared
    # colour tolerance: 1e-8, or -- wild magnitudes -- 32 x the fp32 oracle's own distance from the float64 oracle: one fp32 evaluation order
    # (torch's blocked CPU sums) is a single sample of the rounding noise; the exact-fp32 MFMA path (sequential 4-wide accumulation, hardware
    # exp2 / log2) was seen at 5 x and 24 x that sample in 2 of 1 500 cases, both with |planes| or |W1| in the hundreds to ten-thousands
    if not (mse < max(1e-8, 32 * floor) and de < 5e-4 and we < max(5e-4, 4 * w_floor)):
        if only is not None:
            bad = (depth.cpu() - ref_depth).abs().flatten().argmax()
            print(json.dumps(dict(case=case, floor=floor, w_floor=w_floor, mse=mse, de=de, we=we, ref_w_min=float(ref_w.min()), ref_w_at=float(ref_w.flatten()[bad]), w_at=float(wsum.cpu().flatten()[bad]),
                                  ref_depth_at=float(ref_depth.flatten()[bad]), depth_at=float(depth.cpu().flatten()[bad]), planes_absmax=float(planes.abs().max()), w1_absmax=float(dec[0].abs().max()))))
        fails.append(dict(case=case, wild=wild, interleaved=interleaved, floor=floor, choice=gnerf_hip.last_mlp_choice(dev), S=S, F=F, N=N, res=res, hw=hw, white_back=white_back, disparity=disparity, per_ray=per_ray, mse=mse, depth=de, wsum=we))
print(json.dumps({'cases': n_cases, 'worst': worst, 'failures': fails}))
sys.exit(1 if fails else 0)
