#!/usr/bin/env python3
"""What shader clock does the chip hold under which load?  gnerf_clock_sample (one wave on a side stream: s_memtime against the 100 MHz
s_memrealtime) next to (a) nothing, (b) back-to-back render kernels of config 2, (c) the fused 3x3 convolution, (d) a streaming copy.
usage: python tools/clock_check.py"""
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
def render(k=12):
    for _ in range(k):
        gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=RES)
x = (torch.randn(4, 128, 512, 512, device=dev) * 0.5).half().contiguous(memory_format=torch.channels_last)
wpk = gnerf_hip.pack_conv3x3_weights(torch.randn(128, 128, 3, 3, device=dev) / 34)
def conv(k=16):
    for _ in range(k): gnerf_hip.conv3x3_epilogue(x, wpk, None, gain=1.4, clamp=256.0)
big = torch.empty(1 << 28, device=dev, dtype=torch.float16)
def copy(k=40):
    for _ in range(k): big[: 1 << 27].copy_(big[1 << 27:])
def idle(): pass
rows = {}
for name, fn in (('idle', idle), ('render kernels back to back', render), ('fused 3x3 convolution back to back', conv), ('streaming copy', copy)):
    fn(); torch.cuda.synchronize()
    vals = [gnerf_hip.clock_under_load(fn, microseconds=3000.0, device=dev) for _ in range(5)]
    rows[name] = [round(v, 1) for v in vals]
print(json.dumps({'mhz': rows, 'note': 'five samples of 3 ms each; the load is enqueued before and after the sampler starts'}))
