#!/usr/bin/env python3
"""Render-kernel time for a few (items, res, S+F) shapes, incl. gen_videos.py's doubled 96+96 sampling."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import bench, gnerf_hip
dev = torch.device('cuda', 0)
for (N, RES, S, F) in [(4, 128, 48, 48), (4, 128, 96, 96), (1, 64, 96, 96), (1, 64, 48, 48), (8, 64, 96, 96), (1, 128, 96, 96), (4, 128, 64, 64), (4, 128, 128, 128)]:
    planes, dec, c2w, intr = bench._scene(dev, 1000, n_items=N)
    o, d = gnerf_hip.make_rays(c2w, intr, RES)
    nhwc = gnerf_hip.planes_to_nhwc(planes)
    nc = torch.rand(N * RES * RES, S, device=dev); nf = torch.rand(N * RES * RES, F, device=dev)
    run = lambda: gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=RES)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    rays = N * RES * RES
    print(json.dumps({'items': N, 'res': RES, 'samples': f'{S}+{F}', 'ms': round(best, 4), 'Mrays_s': round(rays / best / 1e3, 2), 'Msamples_s': round(rays * (S + F) / best / 1e3, 1)}))
