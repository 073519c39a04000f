#!/usr/bin/env python3
"""Config 3 (full generator forward, batch 4) and the generator forward of config 5's training shape for rocprofv3, with the warm-up (library
load, MIOpen's solver search) EXCLUDED: the timed passes sit between two marker kernels, tools/prof_orbit.sh keeps what is between them
(MARKED_SCRIPT=tools/config3_marked.py bash tools/prof_orbit.sh <tag> [--part backbone|synthesis] [--fp32] [--reps 10]).  "frames" = images."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
import gnerf_hip, gnerf_harness as H, gnerf_generator

ap = argparse.ArgumentParser()
ap.add_argument('--part', default='synthesis', help='synthesis: backbone + renderer + superresolution; backbone: the plane image only')
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--fp32', action='store_true', help='force_fp32: the superresolution in float32 too')
args = ap.parse_args()
dev = torch.device('cuda', 0)
H.configure_backend()
torch.manual_seed(0)
G = gnerf_generator.Generator().eval().requires_grad_(False).to(dev)
with torch.no_grad():
    for n, p in G.named_parameters():
        if n.endswith('noise_strength') or n.endswith('.bias'):
            p.add_(torch.randn_like(p) * 0.1)
    z = torch.randn(args.batch, G.z_dim, device=dev)
    c = torch.cat([H.camera_label(H.orbit_pose(3 + 7 * i, 120)) for i in range(args.batch)]).to(dev)
    ws = G.mapping(z, c)
    kw = dict(force_fp32=True) if args.fp32 else {}

    def run():
        if args.part == 'backbone':
            return G.backbone.synthesis(ws[:, :G.backbone.num_ws], noise_mode='const')
        return G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=64, **kw)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    gnerf_hip.torch_rand(424242, dev, 1, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gnerf_hip.torch_rand(424242, dev, 1, 0)
    torch.cuda.synchronize()
print(json.dumps({'part': args.part, 'batch': args.batch, 'frames': args.batch * args.reps, 'fp32': bool(args.fp32), 'f32x3': gnerf_generator._F32X3,
                  'ms_per_batch_under_profiler': 1e3 * dt / args.reps, 'frames_per_s_under_profiler': args.batch * args.reps / dt}))
