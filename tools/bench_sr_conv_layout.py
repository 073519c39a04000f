#!/usr/bin/env python3
"""fp16 convolutions of the superresolution blocks (batch 4) in NCHW and in channels_last memory format: what MIOpen's choice costs
per layout (the NCHW calls include its own layout transposes).  One JSON line per layer."""
import json, torch
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
import sys
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, cin, cout, R, up in (('block0.conv0', 32, 256, 128, 2), ('block0.conv1', 256, 256, 256, 1), ('block1.conv0', 256, 128, 256, 2), ('block1.conv1', 128, 128, 512, 1),
                              ('block1.torgb', 128, 3, 512, 0), ('block0.torgb', 256, 3, 256, 0)):
    x = torch.randn(N, cin, R, R, device=dev, dtype=torch.float16)
    k = 1 if up == 0 else 3
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).half()
    out = {'layer': name, 'in': [N, cin, R, R], 'out_ch': cout}
    for fmt_name, fmt in (('nchw', torch.contiguous_format), ('channels_last', torch.channels_last)):
        xf = x.contiguous(memory_format=fmt)
        if up == 2:
            wt = w.transpose(0, 1).contiguous(memory_format=fmt)
            fn = lambda: F.conv_transpose2d(xf, wt, stride=2)
        else:
            wf = w.contiguous(memory_format=fmt)
            fn = lambda: F.conv2d(xf, wf, padding=k // 2)
        y = fn()
        out[fmt_name + '_us'] = timeit(fn)
        out[fmt_name + '_out_is_channels_last'] = bool(y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous())
    print(json.dumps(out), flush=True)
