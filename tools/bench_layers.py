#!/usr/bin/env python3
"""Per-layer time of the config-3 generator forward (batch 4): every StyledConv / ToRGB call with its input shape and dtype,
timed with HIP events (eager, one event pair per call), sorted by time.  python tools/bench_layers.py"""
import json, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
import gnerf_harness as H, gnerf_generator
dev = torch.device('cuda', 0)
torch.manual_seed(0)
G = gnerf_generator.Generator().eval().requires_grad_(False).to(dev)
N = 4
z = torch.randn(N, 512, device=dev)
c = torch.cat([H.camera_label(H.orbit_pose(7 * i, 240)) for i in range(N)]).to(dev)
records = collections.OrderedDict()
def wrap(name, mod):
    orig = mod.forward
    def fwd(x, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = orig(x, *a, **k); e1.record()
        y0 = y[0] if isinstance(y, tuple) else y
        records.setdefault(name, {'in': list(x.shape), 'dtype': str(x.dtype).replace('torch.', ''), 'out': list(y0.shape), 'fmt': 'channels_last' if (x.ndim == 4 and x.stride(1) == 1 and x.shape[1] > 1) else 'nchw', 'ev': []})['ev'].append((e0, e1))
        return y
    mod.forward = fwd
for name, mod in G.named_modules():
    if isinstance(mod, (gnerf_generator.StyledConv, gnerf_generator.ToRGB)):
        wrap(name, mod)
with torch.no_grad():
    ws = G.mapping(z, c)
    for _ in range(6):
        G.synthesis(ws, c, neural_rendering_resolution=64, noise_mode='const')
    torch.cuda.synchronize()
rows = []
for name, r in records.items():
    t = sorted(a.elapsed_time(b) for a, b in r['ev'][2:])
    rows.append((t[len(t) // 2], name, r['in'], r['dtype'] + ' ' + r['fmt']))
tot = sum(r[0] for r in rows)
for t, name, shp, dt in sorted(rows, reverse=True):
    print(json.dumps({'layer': name, 'ms': round(t, 3), 'in': shp, 'dtype': dt}))
print(json.dumps({'sum_ms': round(tot, 3)}))
