#!/usr/bin/env python3
"""The orbit's frames/s against the number of cameras per synthesis call (bench.py's ORBIT_VIEWS), replayed from a HIP graph and with plain launches:
240 frames, 64 x 64 rays x (96+96) samples, cached backbone, SR to 512^2, uint8 frames -- the secondary metric's workload on one GPU.
usage: python tools/bench_orbit_views.py [k ...]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import gen_videos_mi355x as gv
dev = torch.device('cuda', 0)
ks = [int(a) for a in sys.argv[1:]] or [4, 6, 8, 10, 12, 16]
torch.backends.cudnn.benchmark = True
G = gv.build_random_generator(0, dev)
z = torch.randn(1, G.z_dim, device=dev)
n_frames = 240
gv.render_orbit(G, z, 8, 64, dev, double_depth=False)                      # warm-up: backbone, solver search, per-latent constants
for rnd in range(2):
    for k in ks:
        gv.render_orbit(G, z, k, 64, dev, double_depth=False, frames_per_call=k)
        program = gv.FrameProgram(G, gv.orbit_latents(G, z, dev), 64, dev, batch=k)
        row = {'views_per_call': k}
        for name, prog in (('graph', program), ('eager', None)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            frames, _, _ = gv.render_orbit(G, z, n_frames, 64, dev, double_depth=False, program=prog, frames_per_call=k)
            torch.cuda.synchronize()
            row[name + '_frames_per_s'] = round(n_frames / (time.perf_counter() - t0), 1)
        del program
        print(json.dumps(row), flush=True)
