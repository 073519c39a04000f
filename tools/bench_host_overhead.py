#!/usr/bin/env python3
"""Host-side cost of one custom-op call (python wrapper + binding + launch), measured on tiny tensors where the GPU is idle:
wall time per call over 2000 calls, for the public ops and for the raw bindings.  The public ops use the C++ extension
(gnerf_torch_ext.so) when it is built; run with GNERF_HIP_BINDING=ctypes for the ctypes route."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import bias_act, upfirdn2d
import gnerf_hip
dev = torch.device('cuda', 0)
x = torch.randn(1, 8, 16, 16, device=dev)
b = torch.randn(8, device=dev)
f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)

def wall(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

e = gnerf_hip.ext()
null = torch.empty([0])
with torch.no_grad():
    out = {
        'binding of the public ops': 'ctypes' if e is None else 'gnerf_torch_ext (pybind)',
        'bias_act (public op) us': wall(lambda: bias_act.bias_act(x, b, act='lrelu')),
        'upfirdn2d (public op) us': wall(lambda: upfirdn2d.upfirdn2d(x, f, padding=1)),
        'gnerf_hip.bias_act (binding) us': wall(lambda: gnerf_hip.bias_act(x, b, None, None, None, 0, 1, 3, 0.2, 1.41, -1.0)),
        **({} if e is None else {'gnerf_torch_ext.bias_act (binding) us': wall(lambda: e.bias_act(x, b, null, null, null, 0, 1, 3, 0.2, 1.41, -1.0)),
                                 'gnerf_torch_ext.upfirdn2d (binding) us': wall(lambda: e.upfirdn2d(x, f, 1, 1, 1, 1, 1, 1, 1, 1, False, 1.0))}),
        'torch leaky_relu us': wall(lambda: torch.nn.functional.leaky_relu(x, 0.2)),
        'torch add+leaky_relu+mul us': wall(lambda: torch.nn.functional.leaky_relu(x + b.reshape(1, -1, 1, 1), 0.2) * 1.41),
    }
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))
