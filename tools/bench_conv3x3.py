#!/usr/bin/env python3
"""The fused 3x3 convolution + epilogue (csrc/conv3x3.hip) against MIOpen's convolution followed by gnerf_modconv_epilogue_nhwc, on the
superresolution's shapes (4 frames per call): values and time.   usage: python tools/bench_conv3x3.py [--search 0|1] [--shapes small]"""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import torch.nn.functional as F
import gnerf_hip

ap = argparse.ArgumentParser()
ap.add_argument('--search', type=int, default=1, help='MIOpen solver search (torch.backends.cudnn.benchmark)')
ap.add_argument('--shapes', default='sr')
ap.add_argument('--reps', type=int, default=20)
args = ap.parse_args()
torch.backends.cudnn.benchmark = bool(args.search)
dev = torch.device('cuda', 0)
SHAPES = {'sr': [(4, 128, 128, 512, 512), (4, 256, 256, 256, 256)], 'small': [(2, 128, 128, 16, 64), (1, 256, 128, 8, 32), (3, 128, 256, 24, 96)]}[args.shapes]


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


for (n, cin, cout, h, w) in SHAPES:
    g = torch.Generator(device='cpu').manual_seed(1)
    x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
    w16 = wt.half().contiguous(memory_format=torch.channels_last)
    wpk = gnerf_hip.pack_conv3x3_weights(wt)
    dco = (torch.rand(n, cout, generator=g) + 0.5).to(dev)
    nxt = (torch.rand(n, cout, generator=g) + 0.5).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    noise = (torch.randn(h, w, generator=g) * 0.05).to(dev)
    row = {'shape': [n, cin, cout, h, w], 'GFLOP': 2e-9 * n * h * w * cin * cout * 9}
    for name, kw in (('scale+next', dict(scale=dco, next_scale=nxt)), ('scale+noise', dict(scale=dco, noise=noise, round_noise=True)), ('plain', {})):
        def composed():
            y = F.conv2d(x, w16, padding=1)
            return gnerf_hip.modconv_epilogue(y, bias.half(), act='lrelu', gain=2 ** 0.5, clamp=256.0, **kw)
        def fused():
            return gnerf_hip.conv3x3_epilogue(x, wpk, bias, gain=2 ** 0.5, clamp=256.0, **kw)
        a, b = composed(), fused()
        torch.cuda.synchronize()
        assert b.shape == a.shape and gnerf_hip.is_channels_last(b)
        err = float((a.float() - b.float()).abs().max()); ref = float(a.float().abs().max())
        # against the whole chain in fp32 on the same fp16 operands (the arbiter between the two fp16 results; round 6: the fused form rounds
        # once, at the output, the two-launch form three times on the way)
        y32 = F.conv2d(x.float(), w16.float(), padding=1)
        t32 = y32 * (kw['scale'][:, :, None, None] if 'scale' in kw else 1.0) + (kw['noise'] if 'noise' in kw else 0.0)
        t32 = t32 + bias.half().float()[None, :, None, None]
        ref32 = (torch.nn.functional.leaky_relu(t32, 0.2) * 2 ** 0.5).clamp(-256.0, 256.0) * (kw['next_scale'][:, :, None, None] if 'next_scale' in kw else 1.0)
        row[name] = {'max_abs_diff': err, 'max_abs_ref': ref, 'mismatch_frac_above_2ulp': float(((a.float() - b.float()).abs() > 2 ** -9 * a.float().abs().clamp_min(2 ** -6)).float().mean()),
                     'chain_err_fused_vs_fp32': float((b.float() - ref32).abs().max()), 'chain_err_two_launch_vs_fp32': float((a.float() - ref32).abs().max()),
                     'conv_err_fused_vs_fp32': float((gnerf_hip.conv3x3_epilogue(x, wpk, alpha=1.0).float() - y32).abs().max()),
                     'conv_err_miopen_vs_fp32': float((F.conv2d(x, w16, padding=1).float() - y32).abs().max())}
        del y32, t32, ref32
    t_conv = timeit(lambda: F.conv2d(x, w16, padding=1), args.reps)
    y_tmp = F.conv2d(x, w16, padding=1)
    t_epi = timeit(lambda: gnerf_hip.modconv_epilogue(y_tmp, bias.half(), scale=dco, next_scale=nxt, act='lrelu', gain=2 ** 0.5, clamp=256.0), args.reps)
    t_fused = timeit(lambda: gnerf_hip.conv3x3_epilogue(x, wpk, bias, scale=dco, next_scale=nxt, gain=2 ** 0.5, clamp=256.0), args.reps)
    row.update(miopen_conv_ms=t_conv, epilogue_ms=t_epi, fused_ms=t_fused, fused_PFLOPs=row['GFLOP'] / t_fused * 1e-3, miopen_PFLOPs=row['GFLOP'] / t_conv * 1e-3,
               speedup_vs_conv_alone=t_conv / t_fused, speedup_vs_conv_plus_epilogue=(t_conv + t_epi) / t_fused, miopen_search=bool(args.search))
    print(json.dumps(row))
