#!/usr/bin/env python3
"""fp32 3x3 convolutions of the backbone's shapes (batch 4): NCHW against channels_last, MIOpen solver search on."""
import json, torch, torch.nn.functional as F
dev = torch.device('cuda'); torch.backends.cudnn.benchmark = True
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cl = torch.channels_last
for (n, ci, co, h, up) in [(4, 512, 512, 8, 1), (4, 512, 512, 16, 1), (4, 512, 512, 32, 1), (4, 512, 512, 64, 1), (4, 512, 256, 128, 1), (4, 256, 128, 256, 1),
                           (4, 512, 512, 32, 2), (4, 512, 256, 64, 2), (4, 256, 128, 128, 2)]:
    x = torch.randn(n, ci, h, h, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    res = {'shape': [n, ci, co, h], 'up': up}
    if up == 1:
        res['nchw_us'] = round(timeit(lambda: F.conv2d(x, w, padding=1)), 1)
        xc, wc = x.contiguous(memory_format=cl), w.contiguous(memory_format=cl)
        res['cl_us'] = round(timeit(lambda: F.conv2d(xc, wc, padding=1)), 1)
    else:
        wt = w.transpose(0, 1).contiguous()
        res['nchw_us'] = round(timeit(lambda: F.conv_transpose2d(x, wt, stride=2)), 1)
        xc, wc = x.contiguous(memory_format=cl), wt.contiguous(memory_format=cl)
        res['cl_us'] = round(timeit(lambda: F.conv_transpose2d(xc, wc, stride=2)), 1)
    res['GFLOP'] = round(2 * n * h * h * ci * co * 9 / 1e9, 1)
    print(json.dumps(res), flush=True)
