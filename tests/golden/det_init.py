"""Deterministic parameters, noise and synthetic batches shared by tests/golden/make_golden.py (which drives the REFERENCE's
classes in the build container) and by the tests (which drive this repo's modules, on the CPU and on the GPU box).  Nothing
here depends on either code base: tensors are functions of a name / a call index through numpy's frozen legacy generator
(RandomState: stream-stable across numpy versions), so the fixtures only have to carry outputs, never weights or noise."""

import zlib

import numpy as np
import torch


def det_randn(name, shape, dtype=torch.float32):
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return torch.from_numpy(rs.standard_normal(tuple(shape)).astype(np.float32)).to(dtype)


def det_rand(name, shape):
    """Uniform [0, 1) on torch.rand's own 24-bit grid (k / 2^24: exact in float32, never 1.0)."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
    return torch.from_numpy((rs.randint(0, 1 << 24, size=tuple(shape)).astype(np.float32) / float(1 << 24)))


IMAGE_GAIN = 0.08       # on the ToRGB weights that produce image / image_raw: keeps random-init images inside about [-1, 1]


def det_init_(module, prefix=''):
    """Overwrite every parameter and noise buffer of `module` by a function of its NAME (identical names on both sides are the
    contract, tests/test_generator_cpu.py): tensors keep their initial standard deviation (runtime-scaled weights are N(0,1) /
    lr_multiplier at init); all-constant tensors (biases, noise strengths, the affine layers' ones) get +0.1 N(0,1) so that
    every term of the layer graph is exercised."""
    with torch.no_grad():
        named = list(module.named_parameters()) + [(n, b) for n, b in module.named_buffers() if n.endswith('noise_const')]
        for name, t in named:
            r = det_randn(prefix + name, t.shape)
            std = float(t.float().std()) if t.numel() > 1 else 0.0
            new = r * std if std > 1e-6 else t.detach().float().cpu() + 0.1 * r
            if name.endswith('torgb.weight') and 'superresolution' in name:
                new = new * IMAGE_GAIN
            t.copy_(new.to(t.dtype))
    return module


class DetNoise:
    """While active, torch.randn / torch.rand / torch.rand_like return det_randn / det_rand of the CALL INDEX (the layer graph
    fixes the order: backbone noise layer by layer, then the renderer's two draws), on whatever device was asked for."""

    def __init__(self, tag):
        self.tag, self.calls = tag, 0

    def _next(self, kind, shape, device, dtype):
        name = f'{self.tag}/{self.calls}/{kind}'
        self.calls += 1
        t = det_randn(name, shape) if kind == 'randn' else det_rand(name, shape)
        return t.to(device=device, dtype=dtype or torch.float32)

    @staticmethod
    def _shape(a):
        return tuple(a[0]) if len(a) == 1 and isinstance(a[0], (list, tuple, torch.Size)) else tuple(a)

    def __enter__(self):
        self._saved = (torch.randn, torch.rand, torch.rand_like)
        torch.randn = lambda *a, device=None, dtype=None, **k: self._next('randn', self._shape(a), device, dtype)
        torch.rand = lambda *a, device=None, dtype=None, **k: self._next('rand', self._shape(a), device, dtype)
        torch.rand_like = lambda t, **k: self._next('rand', t.shape, t.device, t.dtype)
        return self

    def __exit__(self, *exc):
        torch.randn, torch.rand, torch.rand_like = self._saved


def orbit_label(i, n=120, radius=2.7):
    """Camera label [1, 25] of orbit frame i (gen_videos.py:155-170 with camera_utils.py:89-106,155-174), in float64 numpy so
    that neither side's camera code is involved."""
    yaw = 3.14 / 2 + 0.7 * np.sin(2 * 3.14 * i / n)
    pitch = 3.14 / 2 - 0.05 + 0.3 * np.cos(2 * 3.14 * i / n)
    pitch = min(max(pitch, 1e-5), np.pi - 1e-5)
    org = np.array([radius * np.sin(pitch) * np.cos(np.pi - yaw), radius * np.cos(pitch), radius * np.sin(pitch) * np.sin(np.pi - yaw)])
    fwd = -org / np.linalg.norm(org)
    up = np.array([0.0, 1.0, 0.0])
    right = -np.cross(up, fwd)
    right /= np.linalg.norm(right)
    up2 = np.cross(fwd, right)
    up2 /= np.linalg.norm(up2)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up2, fwd, org
    k = np.array([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]])
    return torch.from_numpy(np.concatenate([m.reshape(-1), k.reshape(-1)])[None].astype(np.float32))


def synthetic_batch(n=4):
    """The per-GPU batch of BASELINE config 5 (dataset.py:1036-1045 shapes): identity latent z (the encoder is
    out of scope), loss / condition camera labels, the 512x512 loss image in [-1, 1], the 64x64 real depth image, factor 1."""
    c = torch.cat([orbit_label(7 + 23 * i) for i in range(n)])
    cond_c = torch.cat([orbit_label(3 + 31 * i) for i in range(n)])
    return dict(z=det_randn('batch/z', (n, 512)), c=c, condition_c=cond_c,
                loss_image=det_randn('batch/loss_image', (n, 3, 512, 512)).mul(0.4).clamp(-1, 1),
                depth_image=det_rand('batch/depth', (n, 1, 64, 64)).mul(1.05).add(2.25), factor=torch.ones(n))
