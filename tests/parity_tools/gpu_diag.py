#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP renderer with the golden vectors + a rough timing of config 2.
Debug aid for the GPU box (prints, never asserts)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT, os.path.join(ROOT, 'tests')]
import gnerf_hip
from oracle import render_ref as R

dev = torch.device('cuda', 0)
print(torch.cuda.get_device_name(0), gnerf_hip.load().gnerf_build_info().decode())


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)


for case in ['render_nofine.npz', 'render_s12.npz', 'render_s48.npz', 'render_misc.npz']:
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', case)))
    dec = [x.to(dev) for x in R.fold_decoder(*[torch.from_numpy(g[k]) for k in ('w1', 'b1', 'w2', 'b2')], float(g['lr_mul']))]
    N, M = g['out_rgb'].shape[:2]
    S, F = int(g['depth_resolution']), int(g['depth_resolution_importance'])
    nhwc = gnerf_hip.planes_to_nhwc(t(g['planes']))
    out = gnerf_hip.render_forward(nhwc, N, dec, t(g['ray_origins']), t(g['ray_dirs']), t(g['noise_coarse']), t(g['noise_fine']) if F else None,
                                   depth_resolution=S, depth_resolution_importance=F, ray_start=float(g['ray_start']), ray_end=float(g['ray_end']),
                                   box_warp=float(g['box_warp']), white_back=bool(g['white_back']), disparity_space_sampling=bool(g['disparity']),
                                   image_width=int(g['res']), debug=True)
    torch.cuda.synchronize()
    rgb, depth, wsum, dbg = [x.cpu().numpy() for x in out]
    dbg = dbg.reshape(N, M, 8, S + F)
    print(f'--- {case}: S={S} F={F}')
    def err(name, a, b):
        print(f'   {name:16s} max|err| {np.abs(a - b).max():.3e}   (ref max {np.abs(b).max():.3e})')
    err('depths_coarse', dbg[:, :, 0, :S], g['depths_coarse'])
    err('sigma_coarse', dbg[:, :, 1, :S], g['sigma_coarse'])
    if F:
        err('weights_coarse', dbg[:, :, 2, :S - 1], g['weights_coarse'])
        err('depths_fine', dbg[:, :, 3, :F], g['depths_fine'])
        err('sigma_fine', dbg[:, :, 4, :F], g['sigma_fine'])
        err('depths_all', dbg[:, :, 5, :], g['depths_all'])
    err('rgb', rgb, g['out_rgb'])
    err('depth', depth, g['out_depth'])
    err('wsum', wsum, g['out_wsum'])
    print(f'   rgb mse {((rgb - g["out_rgb"]) ** 2).mean():.3e}')

# rough timing, config 2
N, res, S, F = 4, 128, 48, 48
gen = torch.Generator().manual_seed(0)
planes = torch.randn(N, 3, 32, 256, 256, generator=gen).to(dev)
dec = [x.to(dev) for x in R.fold_decoder(torch.randn(64, 32, generator=gen), torch.zeros(64), torch.randn(33, 64, generator=gen), torch.zeros(33))]
c2w = torch.cat([R.lookat_pose(3.14 / 2, 3.14 / 2, 2.7)] * N)
intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1)
o, d = gnerf_hip.make_rays(c2w.to(dev), intr.to(dev), res)
nc = torch.rand(N * res * res, S, device=dev)
nf = torch.rand(N * res * res, F, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    nhwc = gnerf_hip.planes_to_nhwc(planes)
    torch.cuda.synchronize(); t1 = time.time()
    for _ in range(5):
        out = gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)
    torch.cuda.synchronize(); t2 = time.time()
    print(f'config2: repack {1e3 * (t1 - t0):.3f} ms, render {1e3 * (t2 - t1) / 5:.3f} ms -> {N * res * res / ((t2 - t1) / 5) / 1e6:.2f} Mrays/s')
print('rgb stats', float(out[0].mean()), float(out[0].std()), 'wsum mean', float(out[2].mean()))
