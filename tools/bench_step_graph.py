#!/usr/bin/env python3
"""The bench step (make_rays + the two torch.rand draws + fused render) launched eagerly and replayed from a captured HIP graph:
what the launch gaps between the step's five kernels cost.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import bench, gnerf_hip
dev = torch.device('cuda', 0)
planes, dec, c2w, intr = bench._scene(dev, 1000)
N, RES, S, F = bench.N_ITEMS, bench.RES, bench.S_COARSE, bench.S_FINE
planes_cl = planes.reshape(N, 96, bench.PLANE, bench.PLANE).permute(0, 2, 3, 1).contiguous()
amax = gnerf_hip.planes_absmax(planes_cl)

def step():
    o, d = gnerf_hip.make_rays(c2w, intr, RES)
    nc = torch.rand([N, RES * RES, S, 1], device=dev)
    nf = torch.rand(N * RES * RES, F, device=dev)
    return gnerf_hip.render_forward(planes_cl, N, dec, o, d, nc, nf, depth_resolution=S, depth_resolution_importance=F, ray_start=bench.RAY_START,
                                    ray_end=bench.RAY_END, box_warp=bench.BOX_WARP, image_width=RES, planes_absmax=amax)

def timed(fn, k=50, reps=5):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): fn()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / k * 1e3)
    return sorted(out)[len(out) // 2]

with torch.no_grad():
    for _ in range(400): step()                       # clocks settle
    eager = timed(step)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    a = out[0].clone(); g.replay(); torch.cuda.synchronize()
    fresh_noise = not torch.equal(a, out[0])          # every replay draws new noise (the generator's offset advances)
    graph = timed(g.replay)
rays = N * RES * RES
print(json.dumps({'eager_ms_per_step': round(eager, 4), 'graph_ms_per_step': round(graph, 4), 'eager_Mrays_s': round(rays / eager / 1e3, 1),
                  'graph_Mrays_s': round(rays / graph / 1e3, 1), 'replays_draw_fresh_noise': fresh_noise}))
