#!/usr/bin/env python3
"""Run the -DGNERF_STAMPS diagnostic build of the pipelined render kernel on config 2 and print where the
shader waves and the scalar wave spend their cycles (shares; the run time of this build is not meaningful).
usage: GNERF_HIP_LIB=g-nerf_amd/gnerf_hip/variants/libgnerf_STAMPS.so python tools/stamps.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import numpy as np
import torch
import bench, gnerf_hip
dev = torch.device('cuda', 0)
planes, dec, c2w, intr = bench._scene(dev, 1000)
N, RES, S, F = bench.N_ITEMS, bench.RES, bench.S_COARSE, bench.S_FINE
if os.environ.get('STAMPS_SAMPLES'):
    S = F = int(os.environ['STAMPS_SAMPLES'])
o, d = gnerf_hip.make_rays(c2w, intr, RES)
nhwc = gnerf_hip.planes_to_nhwc(planes)
nc = torch.rand(N * RES * RES, S, device=dev); nf = torch.rand(N * RES * RES, F, device=dev)
os.environ['GNERF_RENDER_KERNEL'] = 'pipe'
for _ in range(3):
    out = gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=RES, debug=True)
torch.cuda.synchronize()
dbg = out[3].cpu().numpy().view(np.uint64).reshape(-1)
G = 1024 if S <= 48 else (768 if S <= 96 else 512)          # resident workgroups of pipe<1> / pipe<2> / pipe<3> (4, 3, 2 per CU)
st = dbg[:G * 4 * 16].reshape(G, 4, 16).astype(np.float64)
names = {0: 'slot params', 1: 'tap setup', 2: 'lookups', 3: 'layer1', 4: 'act+layer2', 5: 'step tail', 6: 'barrier(even)', 7: 'barrier(odd)',
         8: 'colour weights (v_e)', 13: 'merge ranks', 14: 'final march', 9: 'outputs', 10: 'colour acc', 11: 'coarse march+importance', 12: 'depth proposals'}
nr = st[:, 0, 15].mean()
for role, sel in (('shader waves', st[:, :3].reshape(-1, 16)), ('scalar wave', st[:, 3])):
    tot = sel[:, :15].sum(1).mean()
    print(f'{role}: {tot:.0f} cycles per workgroup run of {nr:.1f} rays = {tot / nr:.0f} cycles/ray')
    for i in range(15):
        v = sel[:, i].mean()
        if v > 0:
            print(f'   {names.get(i, i):28s} {v / nr:9.0f} cyc/ray  {100 * v / tot:5.1f} %')

# where the waves sit and what clock the kernel ran at (GNERF_STAMPS builds write these behind the stamp table)
ext = dbg[G * 4 * 16:G * 4 * 16 + G * 4 * 4].reshape(G, 4, 4)
hw = ext[:, :, 0].astype(np.int64)
simd = (hw >> 4) & 3
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (((hw >> 32) & 15) << 8)
import collections
per_cu = collections.defaultdict(list)
for wg in range(G):
    per_cu[int(cu[wg, 0])].append(int(simd[wg, 3]))
print('SIMDs hosting the scalar waves of a CU\'s workgroups (count of CUs):', dict(collections.Counter(tuple(sorted(v)) for v in per_cu.values()).most_common(8)))
print('SIMD of waves 0..3 (count of workgroups):', dict(collections.Counter(''.join(map(str, r)) for r in simd.tolist()).most_common(8)))
cyc, ticks = ext[:, :, 1].astype(np.float64), ext[:, :, 2].astype(np.float64)
ok = ticks > 0
print(f'in-kernel clock: {np.median(cyc[ok] / ticks[ok]) * 100:.0f} MHz median over waves (s_memtime / s_memrealtime x 100 MHz), {cyc[ok].mean():.0f} cycles per wave run')
start = ext[:, 0, 3].astype(np.float64)
print(f'workgroup start spread: {(start.max() - start.min()) / 100:.1f} us; first-to-last end: {((start + ticks[:, 0]).max() - start.min()) / 100:.1f} us')
