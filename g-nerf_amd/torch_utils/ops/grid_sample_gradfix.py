"""Bilinear 2-D `grid_sample` whose gradient is itself differentiable w.r.t. the incoming gradient.

Interface of the reference's torch_utils/ops/grid_sample_gradfix.py (module flag `enabled` :24,
`grid_sample(input, grid)` :28; fixed to mode='bilinear', padding_mode='zeros', align_corners=False).
G-NeRF never calls it: its only user, the ADA augment pipe, is never constructed, and the renderer's lookups
happen inside the fused render kernel on the GPU path.  Kept API-complete.  With `enabled` set, GPU tensors go to the
hand-written gfx950 sampler and its adjoint (csrc/grid_sample.hip through gnerf_hip.grid_sample_2d / _backward); CPU tensors,
float64 and anything else the native kernels do not cover take ATen's sampler, as the reference does everywhere.

Design: the sampler is linear in the image, so one autograd node suffices for every order -- its backward
returns (another application of the adjoint sampler, the analytic grid gradient), and the adjoint's own
backward w.r.t. the incoming gradient is the forward sampler again.
"""

import torch

import gnerf_hip

enabled = False     # when False, grid_sample() is exactly torch.nn.functional.grid_sample

_SAMPLER_ARGS = dict(mode='bilinear', padding_mode='zeros', align_corners=False)


def _should_use_custom_op():
    return enabled


def grid_sample(input, grid):
    if not _should_use_custom_op():
        return torch.nn.functional.grid_sample(input=input, grid=grid, **_SAMPLER_ARGS)
    return _Sample.apply(input, grid)


def _native(image, grid):
    # no availability check: a GPU call with the library missing must raise (gnerf_hip.load), never fall back silently
    return gnerf_hip.grid_sample_supported(image, grid)


def _forward(image, grid):
    if _native(image, grid):
        return gnerf_hip.grid_sample_2d(image, grid)
    return torch.nn.functional.grid_sample(input=image, grid=grid, **_SAMPLER_ARGS)


def _adjoint(grad_output, image, grid):
    """(d loss / d image, d loss / d grid) of the sampler: the native adjoint on a GPU, else ATen's (bilinear = 0, zeros = 0)."""
    if _native(image, grid):
        return gnerf_hip.grid_sample_2d_backward(grad_output, image, grid)
    return torch.ops.aten.grid_sampler_2d_backward(grad_output, image, grid, 0, 0, False, [True, True])


class _Sample(torch.autograd.Function):
    """image, grid -> samples."""

    @staticmethod
    def forward(ctx, image, grid):
        if image.ndim != 4 or grid.ndim != 4:
            raise AssertionError('grid_sample_gradfix expects a 4-D image and a 4-D grid')
        ctx.save_for_backward(image, grid)
        return _forward(image, grid)

    @staticmethod
    def backward(ctx, grad_samples):
        image, grid = ctx.saved_tensors
        return _SampleAdjoint.apply(grad_samples, image, grid)


class _SampleAdjoint(torch.autograd.Function):
    """grad_samples, image, grid -> (grad_image, grad_grid); differentiable in grad_samples only."""

    @staticmethod
    def forward(ctx, grad_samples, image, grid):
        ctx.save_for_backward(grid)
        return _adjoint(grad_samples, image, grid)

    @staticmethod
    def backward(ctx, gg_image, gg_grid):
        (grid,) = ctx.saved_tensors
        if ctx.needs_input_grad[2]:
            raise AssertionError('second-order gradients w.r.t. the grid are not supported')
        # grad_image is linear in grad_samples with the sampler as its transpose; gg_grid's path is dropped like upstream
        back = _Sample.apply(gg_image, grid) if ctx.needs_input_grad[0] else None
        return back, None, None
