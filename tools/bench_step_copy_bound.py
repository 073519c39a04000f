#!/usr/bin/env python3
"""How far is the headline step's NCHW -> NHWC repack from a plain copy of the same bytes in the same cache state?  The step (repack + rays and draws +
render) timed with the repack as shipped, with the repack replaced by a same-size device copy (torch's copy kernel: the render kernel then reads
planes repacked once, before the loop), and with nothing in its place.   usage: python tools/bench_step_copy_bound.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import bench, gnerf_hip
dev = torch.device('cuda', 0)
planes, dec, c2w, intr = bench._scene(dev, 1000)
N, RES, S, F = bench.N_ITEMS, bench.RES, bench.S_COARSE, bench.S_FINE
nhwc0, amax0 = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
dummy = torch.empty_like(planes)


def step(mode):
    if mode == 'repack':
        nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
    else:
        if mode == 'copy':
            dummy.copy_(planes)
        nhwc, amax = nhwc0, amax0
    o, d, nc, nf = gnerf_hip.make_rays_and_draws(c2w, intr, RES, S, F)
    return gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0,
                                    image_width=RES, planes_absmax=amax)


res = {}
for rnd in range(3):
    for mode in ('repack', 'copy', 'none'):
        for _ in range(20): step(mode)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): step(mode)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(mode, []).append(round(e0.elapsed_time(e1) / 100, 4))
print(json.dumps({'ms_per_step': res, 'repack_minus_none_us': round(1e3 * (min(res['repack']) - min(res['none'])), 1), 'copy_minus_none_us': round(1e3 * (min(res['copy']) - min(res['none'])), 1),
                  'note': 'copy = torch copy kernel of the same 100 MB to a 100 MB buffer in place of the repack (read once + write once, the same bytes)'}))
