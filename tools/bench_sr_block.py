#!/usr/bin/env python3
"""What the custom ops buy inside one superresolution synthesis layer (networks_stylegan2.py:280-334 with up=2 at the
256 -> 512 block, 128 channels, fp16, batch 4 -- the largest layer of a G-NeRF forward): the layer's op sequence
   conv_transpose2d(stride 2)  ->  upfirdn2d 4x4 blur (gain 4)  ->  + noise  ->  bias_act(lrelu, gain sqrt2, clamp 256)
run with the native ops (impl='cuda') and with their PyTorch-op forms (impl='ref', what a G-NeRF user gets without the
plugins).  The convolution is MIOpen's in both cases.  Prints ms per layer call and the share of each part."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import bias_act, upfirdn2d
dev = torch.device('cuda', 0)
N, C, R = 4, 128, 256
dt = torch.float16
x = torch.randn(N, C, R, R, device=dev, dtype=dt)
w = (torch.randn(C, C, 3, 3, device=dev) * 0.05).to(dt)            # conv_transpose2d weight [in, out, kh, kw]
b = torch.randn(C, device=dev, dtype=dt)
noise = torch.randn(1, 1, 2 * R, 2 * R, device=dev, dtype=dt)
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best

def layer(impl):
    y = torch.nn.functional.conv_transpose2d(x, w, stride=2)                               # [N,C,513,513]
    y = upfirdn2d.upfirdn2d(y, f, padding=[1, 1, 1, 1], gain=4, impl=impl)                    # conv2d_resample.py:114-131
    y = y.add_(noise)
    return bias_act.bias_act(y, b, act='lrelu', gain=2 ** 0.5, clamp=256, impl=impl)

with torch.no_grad():
    yc = torch.nn.functional.conv_transpose2d(x, w, stride=2)
    out = {'conv_transpose2d_ms': timeit(lambda: torch.nn.functional.conv_transpose2d(x, w, stride=2))}
    for impl in ('cuda', 'ref'):
        out[f'blur_{impl}_ms'] = timeit(lambda: upfirdn2d.upfirdn2d(yc, f, padding=[1, 1, 1, 1], gain=4, impl=impl))
        yb = upfirdn2d.upfirdn2d(yc, f, padding=[1, 1, 1, 1], gain=4, impl='cuda')
        out[f'bias_act_{impl}_ms'] = timeit(lambda: bias_act.bias_act(yb, b, act='lrelu', gain=2 ** 0.5, clamp=256, impl=impl))
        out[f'layer_{impl}_ms'] = timeit(lambda: layer(impl))
    a, r = layer('cuda').float(), layer('ref').float()
    out['max_abs_diff_cuda_vs_ref'] = float((a - r).abs().max())
print(json.dumps({k: round(v, 4) for k, v in out.items()}))
