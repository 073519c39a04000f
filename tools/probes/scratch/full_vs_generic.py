#!/usr/bin/env python3
"""Diagnostic: render_kernel_pipe<.., FULL=true> against FULL=false (GNERF_PIPE_FULL=0) and against the debug route, per output
and per decoder arithmetic.  Prints one JSON line per comparison."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT, os.path.join(ROOT, 'tests')]
import torch
import gnerf_hip
from test_gpu_parity import _random_scene as random_scene

dev = torch.device('cuda', 0)
for S in (48, 96):
    planes, dec, o, d, nc, nf = random_scene(3, N=2, res=8, S=S, F=S, hw=(16, 16))
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    args = (nhwc, 2, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev))
    kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=8)
    for mlp in ('f16x3', 'f32'):
        os.environ.pop('GNERF_PIPE_FULL', None)
        full = gnerf_hip.render_forward(*args, mlp=mlp, **kw)
        full2 = gnerf_hip.render_forward(*args, mlp=mlp, **kw)
        os.environ['GNERF_PIPE_FULL'] = '0'
        gen = gnerf_hip.render_forward(*args, mlp=mlp, **kw)
        dbg = gnerf_hip.render_forward(*args, mlp=mlp, debug=True, **kw)
        for name, a, b in (('full_vs_full', full, full2), ('full_vs_generic', full, gen), ('generic_vs_debug', gen, dbg[:3])):
            rec = {'S': S, 'mlp': mlp, 'cmp': name}
            for k, x, y in zip(('rgb', 'depth', 'wsum'), a, b):
                diff = (x - y).abs()
                rec[k] = {'n_diff': int((diff > 0).sum()), 'max': float(diff.max()), 'rays': int((diff.reshape(diff.shape[0] * diff.shape[1], -1).amax(1) > 0).sum())}
            print(json.dumps(rec), flush=True)
