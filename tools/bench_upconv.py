#!/usr/bin/env python3
"""How should the x2-upsampling modulated convolution (conv2d_resample.py:114-131: transposed conv, stride 2, then the 4x4 blur) be
issued to MIOpen?  Times, for the layers of a forward pass, the transposed convolution alone in four forms:
  grouped   conv_transpose2d with per-sample weights, groups = N          (the reference's fused_modconv form, inference)
  shared    conv_transpose2d with shared weights on style-scaled input     (the un-fused form, networks_stylegan2.py:76-86)
  poly_g    four polyphase conv2d (2x2, 2x1, 1x2, 1x1 taps), grouped       (no col2im; outputs stay de-interleaved)
  poly_s    four polyphase conv2d, shared weights
Usage: python tools/bench_upconv.py"""
import json, os, sys, time
import torch
import torch.nn.functional as F

dev = torch.device('cuda')
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def phases(w):
    """w [G*O, I, 3, 3] correlation-form weights of the transposed convolution's equivalent; returns the four sub-kernels."""
    return w[:, :, 0::2, 0::2], w[:, :, 0::2, 1:2], w[:, :, 1:2, 0::2], w[:, :, 1:2, 1:2]


for (n, ci, co, h, dt) in [(4, 256, 128, 256, torch.float16), (4, 32, 256, 128, torch.float16), (4, 128, 64, 128, torch.float32),
                           (4, 256, 128, 64, torch.float32), (4, 512, 512, 16, torch.float32), (1, 256, 128, 256, torch.float16)]:
    x = torch.randn(n, ci, h, h, device=dev, dtype=dt)
    w = torch.randn(co, ci, 3, 3, device=dev, dtype=dt) * 0.05
    wn = torch.randn(n, co, ci, 3, 3, device=dev, dtype=dt) * 0.05
    xg = x.reshape(1, n * ci, h, h)
    wt_g = wn.transpose(1, 2).reshape(n * ci, co, 3, 3).contiguous()
    wt_s = w.transpose(0, 1).contiguous()
    res = {'shape': [n, ci, co, h, str(dt)]}
    res['grouped'] = timeit(lambda: F.conv_transpose2d(xg, wt_g, stride=2, groups=n))
    res['shared'] = timeit(lambda: F.conv_transpose2d(x, wt_s, stride=2))
    pg = [p.contiguous() for p in phases(wn.reshape(n * co, ci, 3, 3))]
    ps = [p.contiguous() for p in phases(w)]
    pads = [(1, 1), (1, 0), (0, 1), (0, 0)]
    res['poly_g'] = timeit(lambda: [F.conv2d(xg, p, padding=pd, groups=n) for p, pd in zip(pg, pads)])
    res['poly_s'] = timeit(lambda: [F.conv2d(x, p, padding=pd) for p, pd in zip(ps, pads)])
    res['conv3x3_grouped_same_res'] = timeit(lambda: F.conv2d(xg, wn.reshape(n * co, ci, 3, 3), padding=1, groups=n))
    res['conv3x3_shared_same_res'] = timeit(lambda: F.conv2d(x, w, padding=1))
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}), flush=True)
