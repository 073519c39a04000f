#!/usr/bin/env python3
"""fp32 3x3 convolutions of the backbone (batch 4): MIOpen with per-sample weights (grouped form), MIOpen with shared weights, and
im2col + one fp32 GEMM, for the plain and the x2-upsampling (transposed, stride 2) layers.  One JSON line per shape.
python tools/bench_lowres_conv.py [large]"""
import json, os, sys, time
import torch
import torch.nn.functional as F
dev = torch.device('cuda', 0)
torch.manual_seed(0)

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)

N = 4
shapes = [(512, 512, 4), (512, 512, 8), (512, 512, 16)] if len(sys.argv) < 2 else [(512, 512, 32), (512, 512, 64), (256, 256, 128), (128, 128, 256), (512, 256, 64), (256, 128, 128)]
for C, O, R in shapes:
    x = torch.randn(N, C, R, R, device=dev)
    w = torch.randn(O, C, 3, 3, device=dev) / 68
    wg = torch.randn(N * O, C, 3, 3, device=dev) / 68
    w2 = w.reshape(O, C * 9)
    out = {'in_ch': C, 'out_ch': O, 'res': R}
    out['grouped_us'] = timeit(lambda: F.conv2d(x.reshape(1, N * C, R, R), wg, padding=1, groups=N))
    out['shared_us'] = timeit(lambda: F.conv2d(x, w, padding=1))
    def unfold_mm():
        cols = F.unfold(x, 3, padding=1)                    # [N, C*9, R*R]
        return torch.matmul(w2, cols).reshape(N, O, R, R)
    out['unfold_mm_us'] = timeit(unfold_mm)
    out['unfold_mm_err'] = float((unfold_mm() - F.conv2d(x, w, padding=1)).abs().max())
    # x2 upsampling: transposed stride-2 convolution
    wt = w.transpose(0, 1).contiguous()
    wgt = wg.reshape(N, O, C, 3, 3).transpose(1, 2).reshape(N * C, O, 3, 3).contiguous()
    out['up_grouped_us'] = timeit(lambda: F.conv_transpose2d(x.reshape(1, N * C, R, R), wgt, stride=2, groups=N))
    out['up_shared_us'] = timeit(lambda: F.conv_transpose2d(x, wt, stride=2))
    def up_mm():
        cols = torch.matmul(wt.reshape(C, O * 9).t(), x.reshape(N, C, R * R))          # [N, O*9, R*R]
        return F.fold(cols, (2 * R + 1, 2 * R + 1), 3, stride=2)
    out['up_mm_fold_us'] = timeit(up_mm)
    out['up_mm_err'] = float((up_mm() - F.conv_transpose2d(x, wt, stride=2)).abs().max())
    print(json.dumps(out), flush=True)
