"""Small vector helpers with the names and behaviour of the reference's
training/volumetric_rendering/math_utils.py (transform_vectors :26, normalize_vecs :34,
torch_dot :40, get_ray_limits_box :47, linspace :101)."""

import torch


def transform_vectors(matrix: torch.Tensor, vectors4: torch.Tensor) -> torch.Tensor:
    """Apply an MxM matrix to N row vectors [N,M] (result [N,M])."""
    return vectors4 @ matrix.T


def normalize_vecs(vectors: torch.Tensor) -> torch.Tensor:
    """Scale vectors to unit length along the last axis (no epsilon, like the reference)."""
    return vectors / torch.norm(vectors, dim=-1, keepdim=True)


def torch_dot(x: torch.Tensor, y: torch.Tensor):
    return (x * y).sum(-1)


def get_ray_limits_box(rays_o: torch.Tensor, rays_d: torch.Tensor, box_side_length):
    """Slab test of rays against the cube [-L/2, L/2]^3.

    Returns (t_near, t_far), each shaped like rays_o with a trailing 1; rays that miss get
    (-1, -2) as in the reference (math_utils.py:47-98).  The axes are folded in the order
    x, y, z, and a ray is rejected as soon as a near bound exceeds a far bound of the axes seen
    so far -- the same decisions the reference makes, including for rays parallel to an axis.
    """
    lead = rays_o.shape[:-1]
    o = rays_o.detach().reshape(-1, 3)
    d = rays_d.detach().reshape(-1, 3)
    half = box_side_length / 2
    inv = 1 / d
    neg = inv < 0
    lo = torch.full_like(o, -half)
    hi = torch.full_like(o, half)
    near = (torch.where(neg, hi, lo) - o) * inv          # per-axis entry distance
    far = (torch.where(neg, lo, hi) - o) * inv           # per-axis exit distance
    valid = torch.ones(o.shape[0], dtype=torch.bool, device=o.device)
    tmin, tmax = near[:, 0], far[:, 0]
    for axis in (1, 2):
        valid &= ~((tmin > far[:, axis]) | (near[:, axis] > tmax))
        tmin = torch.max(tmin, near[:, axis])
        tmax = torch.min(tmax, far[:, axis])
    tmin = torch.where(valid, tmin, torch.full_like(tmin, -1))
    tmax = torch.where(valid, tmax, torch.full_like(tmax, -2))
    return tmin.reshape(*lead, 1), tmax.reshape(*lead, 1)


def linspace(start: torch.Tensor, stop: torch.Tensor, num: int):
    """numpy-style linspace over tensors: result [num, *start.shape], endpoints included."""
    frac = torch.arange(num, dtype=torch.float32, device=start.device) / (num - 1)
    frac = frac.reshape([num] + [1] * start.ndim)
    return start[None] + frac * (stop - start)[None]
