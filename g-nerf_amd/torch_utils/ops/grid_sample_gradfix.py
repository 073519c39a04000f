"""`grid_sample` replacement that supports gradients of gradients w.r.t. the sampled image, with the
reference's interface (torch_utils/ops/grid_sample_gradfix.py: module flag `enabled` :24,
grid_sample(input, grid) :28).  2-D, bilinear, zero padding, align_corners=False only.

G-NeRF never calls it (its only user, the ADA augment pipe, is never constructed; the renderer calls
torch.nn.functional.grid_sample directly -- and on the GPU path not even that: the lookups happen
inside the fused render kernel).  Kept API-complete; the device work is ATen's grid sampler."""

import torch

enabled = False     # set True to route grid_sample() through the custom autograd functions


def grid_sample(input, grid):
    if _should_use_custom_op():
        return _GridSample2dForward.apply(input, grid)
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def _should_use_custom_op():
    return enabled


class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid):
        assert input.ndim == 4 and grid.ndim == 4
        ctx.save_for_backward(input, grid)
        return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros', align_corners=False)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        return _GridSample2dBackward.apply(grad_output, input, grid)


class _GridSample2dBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid):
        # interpolation 0 = bilinear, padding 0 = zeros, align_corners False
        grad_input, grad_grid = torch.ops.aten.grid_sampler_2d_backward(grad_output, input, grid, 0, 0, False, [True, True])
        ctx.save_for_backward(grid)
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grid, = ctx.saved_tensors
        grad2_grad_output = None
        if ctx.needs_input_grad[0]:
            # d(grad_input)/d(grad_output) is the sampler itself (it is linear in the image)
            grad2_grad_output = _GridSample2dForward.apply(grad2_grad_input, grid)
        assert not ctx.needs_input_grad[2]
        return grad2_grad_output, None, None
