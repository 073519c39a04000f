#!/usr/bin/env python3
"""Fused channels_last blur + epilogue at the orbit's single-image shapes and at batch 4: us per call and GB/s of algorithmic bytes."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import gnerf_hip
from torch_utils.ops import upfirdn2d
dev = torch.device('cuda', 0)
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
shapes = ((1, 64, 513), (1, 128, 257), (4, 64, 513), (4, 128, 257), (4, 128, 513))
if len(sys.argv) > 1:
    shapes = (tuple(int(v) for v in sys.argv[1].split(',')),)
for n, c, h in shapes:
    x = torch.randn(n, c, h, h, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    sc, nx, b = torch.rand(n, c, device=dev) + 0.5, torch.rand(n, c, device=dev) + 0.5, torch.randn(c, device=dev, dtype=torch.float16)
    run = lambda: gnerf_hip.blur_epilogue_channels_last(x, f, [1, 1, 1, 1], blur_gain=4.0, bias=b, scale=sc, act='lrelu', gain=1.41, clamp=256.0, next_scale=nx)
    for _ in range(5):
        y = run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y = run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(json.dumps({'shape': [n, c, h, h], 'us': round(us, 1), 'GBs': round((x.numel() + y.numel()) * 2 / us / 1e3, 1)}), flush=True)
