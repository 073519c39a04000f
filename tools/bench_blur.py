#!/usr/bin/env python3
"""upfirdn2d 4x4 blur / up2 at the superresolution shapes, one dtype: python tools/bench_blur.py f16|f32 [reps]  (profiling target)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import upfirdn2d
dev = torch.device('cuda', 0)
dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == 'f16') else torch.float32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
es = 2 if dt == torch.float16 else 4
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
cases = {
    'blur [4,128,513,513]->512': (torch.randn(4, 128, 513, 513, device=dev, dtype=dt), dict(padding=[1, 1, 1, 1], gain=4)),
    'blur [4,256,257,257]->256': (torch.randn(4, 256, 257, 257, device=dev, dtype=dt), dict(padding=[1, 1, 1, 1], gain=4)),
    'up2 [4,96,128,128]->256': (torch.randn(4, 96, 128, 128, device=dev, dtype=dt), dict(up=2, padding=[2, 1, 2, 1], gain=4)),
    'down2 [4,128,512,512]->256': (torch.randn(4, 128, 512, 512, device=dev, dtype=dt), dict(down=2, padding=[1, 1, 1, 1])),
}
with torch.no_grad():
    for name, (x, kw) in cases.items():
        for _ in range(3):
            y = upfirdn2d.upfirdn2d(x, f, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            y = upfirdn2d.upfirdn2d(x, f, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = (x.numel() + y.numel()) * es
        print(json.dumps({'op': f'upfirdn2d {name} {sys.argv[1] if len(sys.argv) > 1 else "f16"}', 'ms': round(ms, 4), 'GBs': round(nbytes / ms / 1e6, 1), 'frac_of_8TBs': round(nbytes / ms / 1e6 / 8000, 3)}))
