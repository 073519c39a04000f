#!/usr/bin/env python3
"""Gate for an fp32-grade mode of csrc/conv3x3.hip (round 6): the backbone's fp32 3x3 convolutions
(networks_stylegan2.py:41-98,280-345) on MIOpen against the SAME arithmetic the fused renderer's decoder uses -- every product as
hi*hi + lo*hi + hi*lo of f16 splits, fp32 accumulation -- which for a convolution is simply the f16 kernel on three times the input
channels: x' = [hi(x) | lo(x) | hi(x)], w' = [hi(w) | hi(w) | lo(w)].  Times both on the backbone's three hot shapes (batch 4) and
reports the error of each against a float64 convolution.   usage: python tools/bench_conv_f32grade.py [--search 1]"""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import torch.nn.functional as F
import gnerf_hip

ap = argparse.ArgumentParser()
ap.add_argument('--search', type=int, default=1)
ap.add_argument('--reps', type=int, default=10)
args = ap.parse_args()
torch.backends.cudnn.benchmark = bool(args.search)
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda', 0)


def timeit(fn, reps):
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


def split(t):
    hi = t.half()
    return hi, (t - hi.float()).half()


for (n, c, o, h, w) in [(4, 512, 512, 64, 64), (4, 256, 256, 128, 128), (4, 128, 128, 256, 256), (4, 512, 512, 32, 32)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(n, c, h, w, generator=g) * 1.5).to(dev)
    wt = (torch.randn(o, c, 3, 3, generator=g) / (3 * c ** 0.5)).to(dev)
    gflop = 2 * n * h * w * o * c * 9 / 1e9
    t_miopen = timeit(lambda: F.conv2d(x, wt, padding=1), args.reps)
    xcl = x.contiguous(memory_format=torch.channels_last)
    wcl = wt.contiguous(memory_format=torch.channels_last)
    t_miopen_cl = timeit(lambda: F.conv2d(xcl, wcl, padding=1), args.reps)
    # the split operands (in the product this is the previous layer's epilogue / a pack-time constant; timed separately here)
    def make_x3():
        hi, lo = split(x)
        return torch.cat([hi, lo, hi], 1).contiguous(memory_format=torch.channels_last)
    t_split = timeit(make_x3, args.reps)
    x3 = make_x3()
    whi, wlo = split(wt)
    w3 = gnerf_hip.pack_conv3x3_weights(torch.cat([whi.float(), whi.float(), wlo.float()], 1))
    ok = gnerf_hip.conv3x3_epilogue_supported(x3, o)
    res = {'shape': [n, c, o, h, w], 'GFLOP_fp32': round(gflop, 1), 'miopen_nchw_ms': round(t_miopen, 4), 'miopen_nhwc_ms': round(t_miopen_cl, 4),
           'miopen_TFLOPs': round(gflop / min(t_miopen, t_miopen_cl), 1), 'split_torch_ms': round(t_split, 4), 'supported': bool(ok)}
    if ok:
        t_own = timeit(lambda: gnerf_hip.conv3x3_epilogue(x3, w3, alpha=1.0), args.reps)
        y = gnerf_hip.conv3x3_epilogue(x3, w3, alpha=1.0).float()
        res.update(own_f16x3_ms=round(t_own, 4), own_TFLOPs_fp32_equiv=round(gflop / t_own, 1), speedup_vs_miopen=round(min(t_miopen, t_miopen_cl) / t_own, 2))
        # accuracy on a sub-block, against float64 (the own kernel's fp16 OUTPUT rounding dominates here: the fp32-output form is what would ship)
        ref = F.conv2d(x[:1].double(), wt.double(), padding=1)
        e_mi = float((F.conv2d(x[:1], wt, padding=1).double() - ref).abs().max())
        e_own = float((y[:1].double() - ref).abs().max())
        res.update(max_abs_ref=float(ref.abs().max()), err_miopen_fp32=e_mi, err_own_fp16_output=e_own)
    print(json.dumps(res), flush=True)
