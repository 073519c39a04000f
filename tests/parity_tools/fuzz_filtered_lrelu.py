#!/usr/bin/env python3
"""Randomised parity sweep of the fused filtered_lrelu kernel against the numpy oracle: random resampling factors, tap counts
(zero-padded branches included), asymmetric paddings (also negative = cropping), shapes spanning several tiles, flips, clamp /
no clamp, fp32 and fp16, with the gradient (second fused launch reading the signs) against autograd of the PyTorch-op form.
usage: python tests/parity_tools/fuzz_filtered_lrelu.py [n_cases] [seed]"""
import json, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import numpy as np
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import filtered_lrelu
import gnerf_hip
from oracle import ops_ref as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda', 0)
worst = {'f32': 0.0, 'f16': 0.0, 'grad': 0.0}
fails, ran = [], 0
warnings.simplefilter('ignore')
for case in range(n_cases):
    up, down = int(rng.choice([1, 2, 4])), int(rng.choice([1, 2, 4]))
    fut = 1 if up == 1 else int(rng.integers(up, 8 * up + 1))
    fdt = 1 if down == 1 else int(rng.integers(down, 8 * down + 1))
    N, C = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    H, W = int(rng.integers(3, 60)), int(rng.integers(3, 60))
    pad = [int(v) for v in rng.integers(-2, max(fut, fdt) + 2, size=4)]
    cw = W * up + pad[0] + pad[1] - (fut - 1)
    ch = H * up + pad[2] + pad[3] - (fut - 1)
    if cw <= fdt - 1 + down or ch <= fdt - 1 + down:
        continue
    flip, clamp = bool(rng.integers(0, 2)), (None if rng.integers(0, 3) == 0 else float(rng.uniform(0.2, 1.5)))
    gain, slope = float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.0, 0.5))
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    mk = lambda t: None if t == 1 else (lambda f: (f / np.abs(f).sum()).astype(np.float32))(rng.standard_normal(t))
    fu, fd = mk(fut), mk(fdt)
    kw = dict(up=up, down=down, padding=pad, gain=gain, slope=slope, clamp=clamp, flip_filter=flip)
    ref = O.filtered_lrelu(x, fu, fd, b, **kw)
    tfu = None if fu is None else torch.from_numpy(fu).to(dev)
    tfd = None if fd is None else torch.from_numpy(fd).to(dev)
    one = torch.ones([1, 1], device=dev)
    ran += 1
    info = dict(case=case, up=up, fut=fut, down=down, fdt=fdt, shape=[N, C, H, W], pad=pad, flip=flip, clamp=clamp)
    for dt, key, tol in ((torch.float32, 'f32', 3e-5), (torch.float16, 'f16', 6e-3)):
        xt, bt = torch.from_numpy(x).to(dev).to(dt), torch.from_numpy(b).to(dev).to(dt)
        y, so, rc = gnerf_hip.filtered_lrelu(xt, one if tfu is None else tfu, one if tfd is None else tfd, bt, torch.empty([0]), up, down, *pad, 0, 0,
                                             gain, slope, float('inf') if clamp is None else clamp, flip, bool(rng.integers(0, 2)))
        r = ref if dt == torch.float32 else O.filtered_lrelu(xt.float().cpu().numpy(), fu, fd, bt.float().cpu().numpy(), **kw)
        err = float(np.abs(y.float().cpu().numpy() - r).max() / max(1.0, np.abs(r).max())) if rc == 0 else float('inf')
        worst[key] = max(worst[key], err)
        if not (rc == 0 and tuple(y.shape) == r.shape and err < tol):
            fails.append(dict(info, dtype=key, rc=rc, err=err))
    # gradient through the public op (both launches fused) vs autograd of the PyTorch-op form on the CPU
    xg, bg = torch.from_numpy(x).to(dev).requires_grad_(True), torch.from_numpy(b).to(dev).requires_grad_(True)
    y = filtered_lrelu.filtered_lrelu(xg, fu=tfu, fd=tfd, b=bg, **kw)
    gy = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    dx, db = torch.autograd.grad(y, (xg, bg), gy.to(dev))
    xr, br = torch.from_numpy(x).requires_grad_(True), torch.from_numpy(b).requires_grad_(True)
    yr = filtered_lrelu.filtered_lrelu(xr, fu=None if fu is None else torch.from_numpy(fu), fd=None if fd is None else torch.from_numpy(fd), b=br, impl='ref', **kw)
    dxr, dbr = torch.autograd.grad(yr, (xr, br), gy)
    # a sample within rounding of a kink (0 or +-clamp) may take the other branch: compare in a norm that forgives isolated flips
    e = float((dx.cpu() - dxr).abs().mean() / max(1e-6, float(dxr.abs().mean())))
    worst['grad'] = max(worst['grad'], e)
    if not e < 2e-3:
        fails.append(dict(info, dtype='grad', err=e))
print(json.dumps({'cases_run': ran, 'worst_rel_err': worst, 'failures': fails[:10], 'n_failures': len(fails)}))
sys.exit(1 if fails else 0)
