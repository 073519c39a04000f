#!/usr/bin/env python3
"""ONE upfirdn2d shape in a loop (profiling target): python tools/bench_blur_one.py f16|f32 [blur|blur256|up2|down2] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import upfirdn2d
dev = torch.device('cuda', 0)
dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'f16') else torch.float32
which = sys.argv[2] if len(sys.argv) > 2 else 'blur'
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
x, kw = {'blur': (torch.randn(4, 128, 513, 513, device=dev, dtype=dt), dict(padding=[1, 1, 1, 1], gain=4)),
         'blur256': (torch.randn(4, 256, 257, 257, device=dev, dtype=dt), dict(padding=[1, 1, 1, 1], gain=4)),
         'up2': (torch.randn(4, 96, 128, 128, device=dev, dtype=dt), dict(up=2, padding=[2, 1, 2, 1], gain=4)),
         'down2': (torch.randn(4, 128, 512, 512, device=dev, dtype=dt), dict(down=2, padding=[1, 1, 1, 1]))}[which]
with torch.no_grad():
    for _ in range(reps):
        y = upfirdn2d.upfirdn2d(x, f, **kw)
torch.cuda.synchronize()
print('algorithmic MB per launch', (x.numel() + y.numel()) * x.element_size() / 1e6)
