"""Bit-level run-to-run comparison of the render kernel on config 2: renders once with the stage dump, re-renders ten times and
reports every ray whose rgb differs, with the first stage (debug slot) and samples that differ.
usage: [GNERF_HIP_LIB=<.so>] [GNERF_RENDER_KERNEL=pipe|coop|generic] python tools/determinism.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch, bench, gnerf_hip
dev = torch.device('cuda', 0)
planes, dec, c2w, intr = bench._scene(dev, 1000)
N, RES, S, F = bench.N_ITEMS, bench.RES, bench.S_COARSE, bench.S_FINE
o, d = gnerf_hip.make_rays(c2w, intr, RES)
nhwc = gnerf_hip.planes_to_nhwc(planes)
nc = torch.rand(N * RES * RES, S, device=dev); nf = torch.rand(N * RES * RES, F, device=dev)
def run(dbg=False):
    return gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=RES, debug=dbg)
a = run(True)
tot = 0; slots = {}
for it in range(10):
    b = run(True)
    bad = (a[0] != b[0]).any(-1).reshape(-1).nonzero().flatten().tolist()
    tot += len(bad)
    for r in bad:
        for slot in range(8):
            m = (a[3][r][slot] != b[3][r][slot]).nonzero().flatten().tolist()
            if m: slots.setdefault(slot, []).append(m[:4]); break
print(os.environ.get('GNERF_HIP_LIB', 'default'), os.environ.get('GNERF_RENDER_KERNEL', 'auto'), 'mismatching rays over 10 reruns:', tot, 'first differing slot -> sample ids', {k: v[:6] for k, v in slots.items()})
