import os, sys, json
ROOT='/root/repo'
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import upfirdn2d
dev = torch.device('cuda', 0)
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
for dt, es in ((torch.float16, 2), (torch.float32, 4)):
    for name, shape, pad in (('misaligned rows: 513 wide, pad 1', (4, 128, 513, 513), [1, 1, 1, 1]),
                             ('aligned loads: 520 wide, pad 0 (crop 5)', (4, 128, 513, 520), [0, -5, 1, 1]),
                             ('aligned rows, loads off by one element: 520 wide, pad 1 (crop 6)', (4, 128, 513, 520), [1, -6, 1, 1])):
        x = torch.randn(*shape, device=dev, dtype=dt)
        for _ in range(3): y = upfirdn2d.upfirdn2d(x, f, padding=pad, gain=4)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = upfirdn2d.upfirdn2d(x, f, padding=pad, gain=4)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(json.dumps({'case': name, 'dtype': str(dt), 'out': list(y.shape), 'ms': round(ms, 4), 'GBs': round((x.numel() + y.numel()) * es / ms / 1e6, 1)}))
