#!/usr/bin/env python3
"""Diagnostic: which dtype conversions and layout copies does ONE orbit frame launch?  Patches Tensor.to / .float / .half / .contiguous /
.clone and prints (op, shape, dtypes, caller) for every call on a GPU tensor that makes a new tensor, for a frame after warm-up."""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
import gnerf_harness as H, gen_videos_mi355x as GV

dev = torch.device('cuda', 0)
G = GV.build_random_generator(0, dev)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
log = collections.Counter()
on = [False]


def wrap(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **kw):
        out = orig(self, *a, **kw)
        if on[0] and isinstance(out, torch.Tensor) and self.is_cuda and out.data_ptr() != self.data_ptr():
            fr = traceback.extract_stack(limit=3)[0]
            log[(name, tuple(self.shape), str(self.dtype), str(out.dtype), f'{os.path.basename(fr.filename)}:{fr.lineno}')] += 1
        return out
    setattr(torch.Tensor, name, f)


for n in ('to', 'float', 'half', 'contiguous', 'clone', 'repeat', 'expand'):
    wrap(n)
with torch.no_grad():
    z = torch.randn(1, G.z_dim, device=dev)
    ws = GV.orbit_latents(G, z, dev)
    cams = torch.cat([H.camera_label(H.orbit_pose(i, 240)) for i in range(3 * k)]).to(dev)
    for i in range(2):
        out = G.synthesis(ws=ws, c=cams[i * k:(i + 1) * k], noise_mode='const', neural_rendering_resolution=64, cache_backbone=(i == 0), use_cached_backbone=(i > 0))
        H.to_uint8(out['image'])
    torch.cuda.synchronize()
    on[0] = True
    out = G.synthesis(ws=ws, c=cams[2 * k:3 * k], noise_mode='const', neural_rendering_resolution=64, use_cached_backbone=True)
    H.to_uint8(out['image'])
    on[0] = False
for key, n in sorted(log.items(), key=lambda kv: -kv[1]):
    print(n, key)
