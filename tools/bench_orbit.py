#!/usr/bin/env python3
"""Renderer-only orbit at gen_videos.py's settings: one item, 64x64 rays per frame, 96+96 samples (gen_videos.py:127-128
doubles the 48+48 of the training config), 240 cameras on the orbit of gen_videos.py:155-158, planes repacked once
(cached backbone).  A frame is make_rays + two torch.rand draws + the fused render: six small kernels, i.e. launch-bound,
so the per-frame sequence is also captured once into a HIP graph (torch.cuda.CUDAGraph) and replayed.
Prints frames/s eager and graph-replayed.   usage: python tools/bench_orbit.py [res] [S]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import gnerf_hip, gnerf_harness

res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = F = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = torch.device('cuda', 0)
torch.manual_seed(0)
planes = torch.randn(1, 3, 32, 256, 256, device=dev)
dec = [torch.randn(64, 32, device=dev) / 32 ** 0.5, torch.zeros(64, device=dev), torch.randn(33, 64, device=dev) / 8, torch.zeros(33, device=dev)]
intr = torch.tensor([[[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]], device=dev)
nhwc = gnerf_hip.planes_to_nhwc(planes)
n_frames = 240
poses = torch.stack([gnerf_harness.orbit_pose(i, n_frames, radius=2.7).reshape(4, 4) for i in range(n_frames)]).to(dev)
kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)

def frame(c2w):
    o, d = gnerf_hip.make_rays(c2w, intr, res)
    nc = torch.rand([1, res * res, S, 1], device=dev)
    nf = torch.rand(res * res, F, device=dev)
    return gnerf_hip.render_forward(nhwc, 1, dec, o, d, nc, nf, **kw)

def run_eager():
    for i in range(n_frames):
        out = frame(poses[i:i + 1])
    return out

static_pose = poses[0:1].clone()
for _ in range(3):
    frame(static_pose)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    static_out = frame(static_pose)

def run_graph():
    for i in range(n_frames):
        static_pose.copy_(poses[i:i + 1])
        graph.replay()
    return static_out

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best

# the graph must reproduce the eager result for the same pose and the same generator state
torch.manual_seed(7); ref = frame(poses[5:6])[0].clone()
static_pose.copy_(poses[5:6]); torch.manual_seed(7); graph.replay(); torch.cuda.synchronize()
same = bool(torch.equal(ref, static_out[0]))
te, tg = timeit(run_eager), timeit(run_graph)
print(json.dumps({'workload': f'renderer-only orbit, {n_frames} frames, {res}x{res} rays x ({S}+{F}) samples, 1 item, cached planes',
                  'eager_frames_per_s': round(n_frames / te, 1), 'graph_frames_per_s': round(n_frames / tg, 1),
                  'eager_us_per_frame': round(te / n_frames * 1e6, 1), 'graph_us_per_frame': round(tg / n_frames * 1e6, 1),
                  'graph_matches_eager_bitwise': same}))
