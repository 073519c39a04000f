"""Soak of the staged backward with each kernel's decoder arithmetic forced (GNERF_BWD_MLP_K1 / _K2) against the all-fp32 pair on varying
gradients: every pair agrees to ~1e-6.  (Until the build removed the packed-fp32 hazard -- profiles/r04_pk_opsel_hazard.md -- K2 = f16x3
was off by ~1e-3 on most launches; this was the reproducer.)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), os.path.join(ROOT, 'tests'), ROOT]
import torch
import gnerf_hip, gnerf_harness as H
dev = torch.device('cuda', 0)
torch.manual_seed(0)
N, res, S = 2, 32, 48
planes = torch.randn(N, 3, 32, 64, 64, device=dev)
dec = [torch.randn(64, 32, device=dev) * 0.18, torch.randn(64, device=dev) * 0.1, torch.randn(33, 64, device=dev) * 0.12, torch.randn(33, device=dev) * 0.1]
c2w = torch.cat([H.lookat_pose(3.14 / 2 + 0.3 * i, 3.14 / 2 - 0.05, 2.7) for i in range(N)]).to(dev)
intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1).to(dev)
o, d = gnerf_hip.make_rays(c2w, intr, res)
M = res * res
nc = torch.rand(N * M, S, device=dev); nf = torch.rand(N * M, S, device=dev)
nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res, planes_absmax=amax)
g_rgb = torch.randn(N, M, 32, device=dev); g_depth = torch.randn(N, M, 1, device=dev); g_w = torch.randn(N, M, 1, device=dev)
def rel(a, b): return float((a - b).abs().max() / b.abs().max())
def run(k1, k2, *g):
    os.environ['GNERF_BWD_MLP_K1'], os.environ['GNERF_BWD_MLP_K2'] = k1, k2
    return gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, *g, **kw)
for k1, k2 in (('f16x3', 'f32'), ('f32', 'f16x3'), ('f16x3', 'f16x3'), ('f32', 'f32')):
    bad, worst, errs = 0, 0.0, []
    for i in range(24):
        sc = [1.0, 1e-6, 3.0][i % 3]
        g = (g_rgb * sc, torch.zeros_like(g_depth) if i % 2 else g_depth * sc, torch.zeros_like(g_w))
        ref = run('f32', 'f32', *g)
        os.environ['GNERF_BWD_KERNEL'] = 'wave'
        wav = run('f32', 'f32', *g)
        os.environ.pop('GNERF_BWD_KERNEL')
        out = run(k1, k2, *g)
        e = rel(out[0], ref[0])
        worst = max(worst, e)
        bad += e > 1e-4
        errs.append(float('%.1e' % e))
    print(json.dumps({'k1': k1, 'k2': k2, 'bad_of_24': bad, 'worst': worst, 'errs': errs}))
