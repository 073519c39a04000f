"""Time the renderer's backward kernel against the forward kernel and against PyTorch autograd through the
PyTorch-op path (what training used before the backward kernel existed).  Usage: python tools/bench_bwd.py [N res]"""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'g-nerf_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
import gnerf_hip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = F = int(sys.argv[3]) if len(sys.argv) > 3 else 48
dev = torch.device('cuda', 0)
torch.manual_seed(0)
planes = torch.randn(N, 3, 32, 256, 256, device=dev)
dec = [torch.randn(64, 32, device=dev) * 0.18, torch.randn(64, device=dev) * 0.1, torch.randn(33, 64, device=dev) * 0.12, torch.randn(33, device=dev) * 0.1]
import numpy as np
import gnerf_harness as H
c2w = torch.cat([H.lookat_pose(3.14 / 2 + 0.3 * i, 3.14 / 2 - 0.05, 2.7) for i in range(N)])
intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1)
o, d = gnerf_hip.make_rays(c2w.to(dev), intr.to(dev), res)
M = res * res
nc = torch.rand(N * M, S, device=dev); nf = torch.rand(N * M, F, device=dev)
g_rgb = torch.randn(N, M, 32, device=dev); g_depth = torch.randn(N, M, 1, device=dev); g_w = torch.randn(N, M, 1, device=dev)
nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
kw = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res)

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

out = {'N': N, 'res': res, 'S': S, 'F': F}
if os.environ.get('BWD_ONLY'):         # profiling aid: only the staged (two-pass) or only the single-pass form, 5 calls
    staged = os.environ['BWD_ONLY'] == 'staged'
    out['only'] = os.environ['BWD_ONLY']
    out['bwd_ms'] = timeit(lambda: gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, staged_scatter=staged, planes_absmax=amax, **kw), 4)
    print(json.dumps(out))
    sys.exit(0)
out['fwd_ms'] = timeit(lambda: gnerf_hip.render_forward(nhwc, N, dec, o, d, nc, nf, **kw))
out['bwd_ms'] = timeit(lambda: gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, planes_absmax=amax, **kw))
out['bwd_single_pass_ms'] = timeit(lambda: gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, staged_scatter=False, **kw))
a = gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, **kw)[0]
b = gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, staged_scatter=False, **kw)[0]
out['staged_vs_single_pass_max_rel'] = float((a - b).abs().max() / b.abs().max())
out['bwd_planes_only_ms'] = timeit(lambda: gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, need_decoder=False, **kw))
out['bwd_decoder_only_ms'] = timeit(lambda: gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, need_planes=False, **kw))
out['memset_ms'] = timeit(lambda: torch.zeros_like(nhwc))

# PyTorch-op path with autograd (reference behaviour)
if os.environ.get('BWD_TORCH', '1') == '1':
    from training.volumetric_rendering.renderer import ImportanceRenderer
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
    class Dec(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.w1, s.b1, s.w2, s.b2 = [torch.nn.Parameter(t.clone()) for t in dec]
        def forward(s, feats, dirs):
            x = feats.mean(1)
            n, m, c = x.shape
            x = x.reshape(n * m, c)
            x = torch.nn.functional.softplus(x @ s.w1.t() + s.b1) @ s.w2.t() + s.b2
            x = x.reshape(n, m, -1)
            return {'rgb': torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001, 'sigma': x[..., 0:1]}
    ren = ImportanceRenderer().to(dev)
    dmod = Dec().to(dev)
    pl = planes.clone().requires_grad_(True)
    opts = dict(depth_resolution=S, depth_resolution_importance=F, ray_start=2.25, ray_end=3.3, box_warp=1.0, clamp_mode='softplus',
                disparity_space_sampling=False, white_back=False)
    def torch_step():
        pl.grad = None
        rgb, depth, w = ren._forward_torch(pl, dmod, o, d, opts)
        ((rgb * g_rgb).sum() + (depth * g_depth).sum() + (w * g_w).sum()).backward()
    def torch_fwd():
        with torch.no_grad():
            ren._forward_torch(pl, dmod, o, d, opts)
    try:
        out['torch_fwd_ms'] = timeit(torch_fwd, 3)
        out['torch_fwd_bwd_ms'] = timeit(torch_step, 3)
        out['torch_peak_GB'] = torch.cuda.max_memory_allocated() / 2**30
    except Exception as e:
        out['torch_error'] = repr(e)[:200]
print(json.dumps(out))
