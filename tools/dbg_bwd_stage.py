"""Does the f16 form of the backward tile kernel (GNERF_BWD_MLP_K2=f16x3) differ from run to run, and where?  Captures the staging
buffer of render_backward (the per-sample dX rows the tile kernel writes) on identical inputs and lists the rows that differ: the
tool that located the packed-fp32 hazard (profiles/r04_pk_opsel_hazard.md; sample 13 of a tile, lanes 48-63).  0 rows since the build
rewrites that instruction form.  (Arguments are labels of repeated passes; the GNERF_BWD_DBG stage flags they once selected are gone.)"""
import os, sys, json
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), os.path.join(ROOT, 'tests'), ROOT]
import torch
import gnerf_hip, gnerf_harness as H
dev = torch.device('cuda', 0)
torch.manual_seed(0)
N, res, S = 2, 32, 48
n_all = 2 * S
planes = torch.randn(N, 3, 32, 64, 64, device=dev)
dec = [torch.randn(64, 32, device=dev) * 0.18, torch.randn(64, device=dev) * 0.1, torch.randn(33, 64, device=dev) * 0.12, torch.randn(33, device=dev) * 0.1]
c2w = torch.cat([H.lookat_pose(3.14 / 2 + 0.3 * i, 3.14 / 2 - 0.05, 2.7) for i in range(N)]).to(dev)
intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]]).repeat(N, 1, 1).to(dev)
o, d = gnerf_hip.make_rays(c2w, intr, res)
M = res * res
nc = torch.rand(N * M, S, device=dev); nf = torch.rand(N * M, S, device=dev)
nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)
kw = dict(depth_resolution=S, depth_resolution_importance=S, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=res, planes_absmax=amax, need_decoder=False)
g_rgb = torch.randn(N, M, 32, device=dev); g_depth = torch.zeros(N, M, 1, device=dev); g_w = torch.zeros(N, M, 1, device=dev)
captured = []
_orig_empty = torch.empty
def _empty(*a, **k):
    t = _orig_empty(*a, **k)
    if k.get('dtype') is torch.uint8: captured.append(t)
    return t
torch.empty = _empty
def stage_rows(k2, flags):
    os.environ['GNERF_BWD_MLP_K1'], os.environ['GNERF_BWD_MLP_K2'], os.environ['GNERF_BWD_DBG'] = 'f32', k2, flags
    del captured[:]
    gnerf_hip.render_backward(nhwc, N, dec, o, d, nc, nf, g_rgb, g_depth, g_w, **kw)
    torch.cuda.synchronize()
    st = captured[-1][:N * M * n_all * 33 * 4].view(torch.float32).view(N * M, n_all * 33).clone()
    return st[:, :n_all].clone(), st[:, n_all:].reshape(N * M, n_all, 32).clone()
for flags in sys.argv[1:] or ['8', '2', '0']:
    runs = [stage_rows('f16x3', flags) for _ in range(12)]
    rows = torch.stack([r[1] for r in runs])                      # [run][ray][sample][32]
    med = rows.median(dim=0).values                               # the value most runs agree on
    bad = (rows != med[None]).any(dim=-1)                         # [run][ray][sample]
    rec = {'flags': flags, 'runs': len(runs), 'bad_rows_per_run': bad.sum(dim=(1, 2)).tolist(), 'depths_differ': int(sum((r[0] != runs[0][0]).sum() for r in runs))}
    print(json.dumps(rec))
    shown = 0
    for idx in bad.nonzero().tolist():
        if shown >= 14: break
        r, ray, smp = idx
        a, b = rows[r, ray, smp], med[ray, smp]
        nz = (a != b).nonzero().flatten().tolist()
        ratio = (a[nz] / b[nz]).tolist()
        # does the wrong row equal another sample's row of the same ray (a stale / misplaced v)?
        same_tile = [s2 for s2 in range(16 * (smp // 16), 16 * (smp // 16) + 16) if s2 != smp and torch.equal(med[ray, s2], a)]
        print(json.dumps({'run': r, 'ray': ray, 'tile': smp // 16, 'smp_in_tile': smp % 16, 'n_ch_differ': len(nz), 'channels': nz[:8],
                          'got': [float('%.4g' % x) for x in a[nz][:4].tolist()], 'want': [float('%.4g' % x) for x in b[nz][:4].tolist()],
                          'ratio': [float('%.5g' % x) for x in ratio[:6]], 'equals_other_sample_of_tile': same_tile,
                          'rows_bad_in_this_tile': int(bad[r, ray, 16 * (smp // 16):16 * (smp // 16) + 16].sum())}))
        shown += 1
