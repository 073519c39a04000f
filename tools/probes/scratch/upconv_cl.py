#!/usr/bin/env python3
"""x2-upsampling convolution of the SR blocks in channels_last fp16: conv_transpose2d (what runs) against four polyphase conv2d."""
import json, torch, torch.nn.functional as F
dev = torch.device("cuda"); torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
cl = torch.channels_last
for (n, ci, co, h) in [(1, 256, 128, 256), (4, 256, 128, 256), (1, 32, 256, 128), (4, 32, 256, 128)]:
    x = torch.randn(n, ci, h, h, device=dev, dtype=torch.float16).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 3, 3, device=dev, dtype=torch.float16) * 0.05)
    wt = w.transpose(0, 1).contiguous(memory_format=cl)                      # [I,O,3,3] as conv_transpose2d takes it
    ref = F.conv_transpose2d(x, wt, stride=2)
    # out[2i+a, 2j+b] = sum over taps of parity (a, b): correlation with the flipped sub-kernels
    wf = w.flip([2, 3])
    subs = {(0, 0): wf[:, :, 0::2, 0::2], (0, 1): wf[:, :, 0::2, 1:2], (1, 0): wf[:, :, 1:2, 0::2], (1, 1): wf[:, :, 1:2, 1:2]}
    subs = {k: v.contiguous(memory_format=cl) for k, v in subs.items()}
    pads = {(0, 0): (1, 1), (0, 1): (1, 0), (1, 0): (0, 1), (1, 1): (0, 0)}
    def poly():
        return {k: F.conv2d(x, subs[k], padding=pads[k]) for k in subs}
    out = poly()
    err = 0.0
    for (a, b), v in out.items():
        err = max(err, float((ref[:, :, a::2, b::2].float() - v.float()).abs().max()))
    res = {'shape': [n, ci, co, h], 'transposed_us': round(timeit(lambda: F.conv_transpose2d(x, wt, stride=2)), 1), 'polyphase_us': round(timeit(poly), 1),
           'same_flops_conv3x3_us': round(timeit(lambda: F.conv2d(x, w.contiguous(memory_format=cl), padding=1)), 1), 'max_abs_diff': err,
           'out_cl': all(v.is_contiguous(memory_format=cl) for v in out.values())}
    print(json.dumps(res), flush=True)
