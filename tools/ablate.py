#!/usr/bin/env python3
"""Time the render kernel of one library build on config 2 (used with tools/build_variants.sh).
usage: GNERF_HIP_LIB=<.so> [GNERF_RENDER_KERNEL=generic] python tools/ablate.py [label]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import bench, gnerf_hip
dev = torch.device('cuda', 0)
planes, dec, c2w, intr = bench._scene(dev, 1000)
N, RES, S, F = bench.N_ITEMS, bench.RES, bench.S_COARSE, bench.S_FINE
o, d = gnerf_hip.make_rays(c2w, intr, RES)
nhwc = gnerf_hip.planes_to_nhwc(planes)
nc = torch.rand(N * RES * RES, S, device=dev); nf = torch.rand(N * RES * RES, F, device=dev)
def run():
    return gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=RES)
for _ in range(5): run()
torch.cuda.synchronize()
reps = 30
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = []
for _ in range(3):
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / reps)
print(json.dumps({'label': sys.argv[1] if len(sys.argv) > 1 else os.environ.get('GNERF_HIP_LIB', 'default'), 'kernel': os.environ.get('GNERF_RENDER_KERNEL', 'auto'), 'ms': min(best), 'all': best}))
