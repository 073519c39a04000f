#!/usr/bin/env python3
"""The phase-decomposed stride-2 transposed 3x3 convolution (csrc/conv3x3.hip, MODE 1) against the framework's conv_transpose2d (MIOpen):
values on small and odd shapes first, then time on the superresolution's x2 layer.   usage: python tools/bench_conv_transpose.py [--time 0|1]"""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import torch.nn.functional as F
import gnerf_hip

ap = argparse.ArgumentParser()
ap.add_argument('--time', type=int, default=1)
ap.add_argument('--search', type=int, default=1)
args = ap.parse_args()
torch.backends.cudnn.benchmark = bool(args.search)
dev = torch.device('cuda', 0)


def case(n, cin, cout, h, w, seed=1):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (2 * cin ** 0.5)).to(dev)
    w16 = wt.half()
    wtT = w16.transpose(0, 1).contiguous(memory_format=torch.channels_last)        # [I, O, 3, 3] as conv_transpose2d wants it
    return x, wt, w16, wtT


rows = []
for shape in [(1, 64, 128, 8, 32), (2, 128, 128, 5, 7), (1, 64, 256, 16, 40), (3, 192, 128, 9, 33), (1, 64, 128, 2, 1)]:
    x, wt, w16, wtT = case(*shape)
    got = gnerf_hip.conv_transpose3x3_s2(x, gnerf_hip.pack_conv_transpose3x3_weights(wt))
    torch.cuda.synchronize()
    want = F.conv_transpose2d(x, wtT, stride=2)
    ref = F.conv_transpose2d(x.float(), w16.float().transpose(0, 1), stride=2)
    n, cin, cout, h, w = shape
    assert got.shape == (n, cout, 2 * h + 1, 2 * w + 1) and got.dtype == torch.float16 and gnerf_hip.is_channels_last(got), (got.shape, got.stride())
    e_got, e_want, top = float((got.float() - ref).abs().max()), float((want.float() - ref).abs().max()), float(ref.abs().max())
    rows.append({'shape': shape, 'err_vs_fp32': e_got, 'miopen_err_vs_fp32': e_want, 'max_abs_ref': top})
    print(json.dumps(rows[-1]), flush=True)
    assert e_got <= max(1.5 * e_want, 2e-3 * top), rows[-1]

if args.time:
    def timeit(fn, reps=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(3):
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
        return min(ts)
    for shape in [(4, 256, 128, 256, 256), (4, 128, 128, 256, 256)]:
        x, wt, w16, wtT = case(*shape)
        wp = gnerf_hip.pack_conv_transpose3x3_weights(wt)
        got = gnerf_hip.conv_transpose3x3_s2(x, wp); want = F.conv_transpose2d(x, wtT, stride=2)
        ref = F.conv_transpose2d(x.float(), w16.float().transpose(0, 1), stride=2)
        n, cin, cout, h, w = shape
        gflop = 2e-9 * n * h * w * cin * cout * 9
        t_mine, t_miopen = timeit(lambda: gnerf_hip.conv_transpose3x3_s2(x, wp)), timeit(lambda: F.conv_transpose2d(x, wtT, stride=2))
        print(json.dumps({'shape': shape, 'GFLOP': gflop, 'fused_ms': t_mine, 'miopen_ms': t_miopen, 'PFLOPs': gflop / t_mine * 1e-3, 'miopen_PFLOPs': gflop / t_miopen * 1e-3,
                          'speedup': t_miopen / t_mine, 'err_vs_fp32': float((got.float() - ref).abs().max()), 'miopen_err_vs_fp32': float((want.float() - ref).abs().max())}), flush=True)
