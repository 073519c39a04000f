"""BASELINE configs 3 and 5 on THIS repo's modules, set up exactly like tests/golden/make_golden.py sets up the reference's
(weights, noise and batch are functions of names / call indices: tests/golden/det_init.py).  Used by the CPU and the GPU tests."""

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import det_init as DI  # noqa: E402


def build(device):
    import gnerf_generator
    torch.manual_seed(0)
    G = DI.det_init_(gnerf_generator.Generator(), 'G/').eval().requires_grad_(False).to(device)
    D = DI.det_init_(gnerf_generator.Discriminator(c_dim=25, img_resolution=64, img_channels=1, mbstd_group_size=4), 'D/')
    return G, D.train().requires_grad_(False).to(device)


def batch_on(device, n=4):
    return {k: v.to(device) for k, v in DI.synthetic_batch(n).items()}


def run_config3(G, device, **synthesis_kwargs):
    b = batch_on(device)
    with torch.no_grad(), DI.DetNoise('config3'):
        ws = G.mapping(b['z'], b['c'])
        out = G.synthesis(ws, b['c'], noise_mode='const', neural_rendering_resolution=64, **synthesis_kwargs)
    return ws, out


def run_config5(G, D, device, force_fp32):
    """One G + D step's forward and backward passes (no exchange, no optimiser): returns (loss terms, images, per-parameter
    gradient norms of G and of D)."""
    import train_step_mi355x as T
    b = batch_on(device)
    kw = dict(force_fp32=True) if force_fp32 else {}
    G.train().requires_grad_(True)
    for p in list(G.parameters()) + list(D.parameters()):
        p.grad = None
    with DI.DetNoise('config5'):
        loss, parts, gen = T.generator_loss(G, lambda img, c: D(img, c, **kw), b, 64, **kw)
        loss.backward()
    g_norms = {n: float(p.grad.double().norm()) for n, p in G.named_parameters() if p.grad is not None}
    G.requires_grad_(False)
    D.requires_grad_(True)
    parts.update(T.discriminator_backward(D, gen['image_depth'], b, **kw))
    D.requires_grad_(False)
    d_norms = {n: float(p.grad.double().norm()) for n, p in D.named_parameters()}
    parts['loss'] = loss.detach()
    G.eval()
    return {k: float(v) for k, v in parts.items()}, gen, g_norms, d_norms


def compare_norms(got, names, want, rel, what):
    """Per-parameter gradient norms against the fixture: relative to each norm, with a floor at 1e-4 of the largest."""
    assert sorted(got) == sorted(str(n) for n in names), f'{what}: parameter names differ from the reference'
    floor = 1e-4 * float(np.max(want))
    worst = max((abs(got[str(n)] - w) / max(w, floor), str(n)) for n, w in zip(names, want))
    assert worst[0] < rel, (what, worst)
    return worst


def compare_whole_gradients(G, D, grads, rel_l2, what):
    """Three whole gradient tensors elementwise against tests/golden/train_step_grads.npz (relative L2 error; the fixture stores
    float16 of value / max|value|: 3e-4 of storage rounding): a sign or permutation error inside a tensor keeps its norm."""
    gp = dict(G.named_parameters())
    got = {'g_backbone_b256_conv1_w': gp['backbone.synthesis.b256.conv1.weight'].grad, 'g_sr_block1_conv0_w': gp['superresolution.block1.conv0.weight'].grad,
           'd_b4_out_w': D.b4.out.weight.grad}
    worst = {}
    for k, t in got.items():
        want = torch.from_numpy(grads[k + '_f16'].astype(np.float32)) * float(grads[k + '_scale'])
        assert tuple(t.shape) == tuple(want.shape), (what, k)
        err = float((t.detach().double().cpu() - want.double()).norm() / want.double().norm())
        assert err < rel_l2, (what, k, err)
        worst[k] = err
    return worst
