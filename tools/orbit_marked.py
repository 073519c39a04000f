#!/usr/bin/env python3
"""The gen_videos orbit (config 4, one GPU) for rocprofv3 with the warm-up EXCLUDED from the statistics: after the warm-up frames
(library load, MIOpen's per-shape solver search -- ~95 % of a plain kernel trace's recorded time) the script
launches a marker kernel (gnerf_torch_rand with numel = 424242, a kernel and a grid no frame uses), runs the orbit, and launches the
marker again; tools/prof_orbit.sh keeps the dispatches between the two markers.
usage: python tools/orbit_marked.py [--flow fast|reference] [--frames 240] [--frames-per-call 1] [--graph]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
import gnerf_hip, gnerf_harness as H, gen_videos_mi355x as gv, gnerf_generator as GG

ap = argparse.ArgumentParser()
ap.add_argument('--flow', default='fast')
ap.add_argument('--frames', type=int, default=240)
ap.add_argument('--frames-per-call', type=int, default=1)
ap.add_argument('--graph', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda', 0)
H.configure_backend()
MARK = 424242


def marker():
    gnerf_hip.torch_rand(MARK, dev, 1, 0)


with torch.no_grad():
    G = gv.build_random_generator(0, dev)
    z = torch.randn(1, G.z_dim, generator=torch.Generator().manual_seed(1)).to(dev)
    fast = args.flow == 'fast'
    GG._MODCONV_FAST = fast
    G.backbone.synthesis.b256.emit_channels_last = fast
    k = args.frames_per_call
    gv.render_orbit(G, z, args.frames, 64, dev, rank=0, world=args.frames, double_depth=True)          # warm-up frame; sets 96+96
    if k > 1:
        gv.render_orbit(G, z, k, 64, dev, double_depth=False, frames_per_call=k)
    program = gv.FrameProgram(G, gv.orbit_latents(G, z, dev), 64, dev) if args.graph else None
    gv.render_orbit(G, z, 8, 64, dev, double_depth=False, program=program, frames_per_call=k)           # a second, short pass: caches, allocator
    torch.cuda.synchronize()
    marker()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    frames, _, _ = gv.render_orbit(G, z, args.frames, 64, dev, double_depth=False, program=program, frames_per_call=k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    marker()
    torch.cuda.synchronize()
print(json.dumps({'flow': args.flow, 'frames': args.frames, 'frames_per_call': k, 'graph': bool(args.graph), 'frames_per_s_under_profiler': args.frames / dt,
                  'ms_per_frame': 1e3 * dt / args.frames}))
