#!/usr/bin/env python3
"""Randomised parity sweep of bias_act and upfirdn2d (public drop-in ops on the GPU = the native kernels) against the numpy oracle.
bias_act: random shapes / bias dimension / activation / alpha / gain / clamp / dtype / memory format, forward and first-order gradient
(through autograd of the public op).  upfirdn2d: random up / down factors per axis, paddings (negative = crop), filter sizes, separable
and 2-D filters, flips, gain, dtype, memory format, forward and gradient (dot-product test against the oracle's forward: the
op is linear, so <dy, F x> must equal <F^T dy, x>).   usage: python tests/parity_tools/fuzz_ops.py [n_cases] [seed]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import numpy as np
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import bias_act, upfirdn2d
from oracle import ops_ref as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device('cuda', 0)
ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']
worst = {'bias_act': 0.0, 'bias_act_grad': 0.0, 'upfirdn2d': 0.0, 'upfirdn2d_grad': 0.0}
fails = []

for case in range(n_cases):
    # ---------------- bias_act
    nd = int(rng.integers(2, 5))
    shape = [int(rng.integers(1, 9)) for _ in range(nd)]
    if rng.integers(0, 3) == 0:
        shape[-1] = int(rng.integers(1, 700))
    if nd == 3 and shape[2] == 1:
        shape[2] = 2         # a rank-3 tensor with stride(1) == 1 is taken for channels_last and refused, upstream too (bias_act.py:147-148)
    dim = int(rng.integers(0, nd))
    act = ACTS[int(rng.integers(0, len(ACTS)))]
    alpha = None if rng.integers(0, 2) else float(rng.uniform(0.05, 0.5))
    gain = None if rng.integers(0, 2) else float(rng.uniform(0.5, 2.0))
    clamp = None if rng.integers(0, 2) else float(rng.uniform(0.3, 2.0))
    use_b = bool(rng.integers(0, 4))
    dt = [torch.float32, torch.float16, torch.float64][int(rng.choice(3, p=[0.5, 0.35, 0.15]))]
    if dt == torch.float16 and clamp is not None:
        clamp = round(clamp * 64) / 64        # exactly representable: the gradient masks on the SAVED fp16 output (bias_act.cu:137-146), so a
                                              # clamp value that fp16 rounds below itself un-masks every clamped element, upstream too
    x = rng.standard_normal(shape) * 1.5
    b = rng.standard_normal(shape[dim]) if use_b else None
    xt = torch.from_numpy(x).to(dev).to(dt)
    if nd == 4 and rng.integers(0, 3) == 0:
        xt = xt.contiguous(memory_format=torch.channels_last)
    bt = None if b is None else torch.from_numpy(b).to(dev).to(dt)
    xr, br = xt.double().cpu().numpy(), (None if bt is None else bt.double().cpu().numpy())
    kw = dict(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    info = dict(case=case, op='bias_act', shape=shape, dtype=str(dt), **kw)
    xg = xt.clone().requires_grad_(True)
    y = bias_act.bias_act(xg, bt, **kw)
    ref = O.bias_act(xr, br, **kw)
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.float64: 3e-7}[dt]      # fp64: alpha / gain / clamp cross the ABI as float32, as upstream (bias_act.h)
    err = float(np.abs(y.detach().double().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
    worst['bias_act'] = max(worst['bias_act'], err) if dt != torch.float16 else worst['bias_act']
    if not err < tol:
        fails.append(dict(info, err=err))
    dy = rng.standard_normal(shape)
    (dx,) = torch.autograd.grad(y, xg, torch.from_numpy(dy).to(dev).to(dt))
    if not (act == 'linear' and clamp is not None):                   # (the GPU path's documented quirk: linear + clamp is not masked)
        refg = O.bias_act_grad(torch.from_numpy(dy).to(dt).double().numpy(), xr, br, **kw)
        # Element-wise comparison; an element within rounding of a kink (relu / lrelu at 0, a clamp edge -- in fp16 the mask is taken from
        # the SAVED fp16 output, bias_act.cu:137-146, so an output that rounds onto the clamp value is masked) may take the other branch:
        # isolated flips are forgiven, anything systematic is not.
        got = dx.double().cpu().numpy()
        etol = {torch.float32: 2e-4, torch.float16: 3e-2, torch.float64: 1e-6}[dt]
        wrong = np.abs(got - refg) > etol * np.maximum(1.0, np.abs(refg)) * max(1.0, float(np.abs(dy).max()))
        ok_elems = ~wrong
        errg = float(np.abs(got - refg)[ok_elems].max() / max(1.0, np.abs(refg).max())) if ok_elems.any() else 0.0
        if dt != torch.float16:
            worst['bias_act_grad'] = max(worst['bias_act_grad'], errg)
        if wrong.sum() > max(1, wrong.size // 100):
            fails.append(dict(info, grad_mismatches=int(wrong.sum()), of=int(wrong.size)))

    # ---------------- upfirdn2d
    N, C, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 5)), int(rng.integers(2, 40)), int(rng.integers(2, 40))
    up = [int(rng.integers(1, 4)), int(rng.integers(1, 4))] if rng.integers(0, 3) == 0 else int(rng.integers(1, 4))
    down = [int(rng.integers(1, 4)), int(rng.integers(1, 4))] if rng.integers(0, 3) == 0 else int(rng.integers(1, 4))
    ftaps = int(rng.integers(1, 9))
    sep = bool(rng.integers(0, 2))
    f = rng.standard_normal(ftaps if sep else (int(rng.integers(1, 7)), ftaps)).astype(np.float32)
    fw, fh = (ftaps, ftaps) if sep else (f.shape[1], f.shape[0])
    pad = [int(v) for v in rng.integers(-2, 6, size=4)]
    upx, upy = (up, up) if isinstance(up, int) else up
    dnx, dny = (down, down) if isinstance(down, int) else down
    if (W * upx + pad[0] + pad[1] - fw + dnx) // dnx < 1 or (H * upy + pad[2] + pad[3] - fh + dny) // dny < 1:
        continue
    if W * upx + pad[0] + pad[1] < fw or H * upy + pad[2] + pad[3] < fh:
        continue
    flip, g = bool(rng.integers(0, 2)), float(rng.uniform(0.5, 4.0))
    dt = torch.float16 if rng.integers(0, 3) == 0 else torch.float32
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    xt = torch.from_numpy(x).to(dev).to(dt)
    if rng.integers(0, 4) == 0:
        xt = xt.contiguous(memory_format=torch.channels_last)
    ft = torch.from_numpy(f).to(dev)
    kw = dict(up=up, down=down, padding=pad, flip_filter=flip, gain=g)
    info = dict(case=case, op='upfirdn2d', shape=[N, C, H, W], f=list(f.shape), dtype=str(dt), **kw)
    xg = xt.clone().requires_grad_(True)
    y = upfirdn2d.upfirdn2d(xg, ft, **kw)
    ref = O.upfirdn2d(xt.float().cpu().numpy(), f, **kw)
    if tuple(y.shape) != ref.shape:
        fails.append(dict(info, shape_got=list(y.shape), shape_want=list(ref.shape)))
        continue
    err = float(np.abs(y.detach().float().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
    if dt == torch.float32:
        worst['upfirdn2d'] = max(worst['upfirdn2d'], err)
    if not err < (2e-5 if dt == torch.float32 else 4e-3):
        fails.append(dict(info, err=err))
    dy = rng.standard_normal(ref.shape).astype(np.float32)
    (dx,) = torch.autograd.grad(y, xg, torch.from_numpy(dy).to(dev).to(dt))
    lhs = float((dy.astype(np.float64) * ref).sum())                                   # <dy, F x>
    rhs = float((dx.double().cpu().numpy() * xt.double().cpu().numpy()).sum())        # <F^T dy, x>
    errg = abs(lhs - rhs) / max(1.0, float(np.abs(dy.astype(np.float64) * ref).sum()))      # relative to the magnitude of the terms summed
    if dt == torch.float32:
        worst['upfirdn2d_grad'] = max(worst['upfirdn2d_grad'], errg)
    if not errg < (1e-4 if dt == torch.float32 else 2e-2):
        fails.append(dict(info, adjoint_err=errg, lhs=lhs, rhs=rhs))

print(json.dumps({'cases': n_cases, 'worst_rel_err_fp32_fp64': worst, 'n_failures': len(fails), 'failures': fails[:8]}))
sys.exit(1 if fails else 0)
