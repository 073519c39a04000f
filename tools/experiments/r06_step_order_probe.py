import os, sys, json
sys.path[:0] = ['g-nerf_amd', '.']
import torch
import bench, gnerf_hip
dev = torch.device('cuda', 0)
planes, dec, c2w, intr = bench._scene(dev, 1000)
N, RES, S, F = bench.N_ITEMS, bench.RES, bench.S_COARSE, bench.S_FINE
def step(order):
    if order == 'repack_first':
        nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
        o, d, nc, nf = gnerf_hip.make_rays_and_draws(c2w, intr, RES, S, F)
    else:
        o, d, nc, nf = gnerf_hip.make_rays_and_draws(c2w, intr, RES, S, F)
        nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
    return gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=RES, planes_absmax=amax)
res = {}
for rnd in range(3):
    for order in ('repack_first', 'draws_first'):
        for _ in range(20): step(order)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): step(order)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(order, []).append(round(e0.elapsed_time(e1) / 100, 4))
print(json.dumps(res))
