"""MipRayMarcher2 with the reference's interface (training/volumetric_rendering/ray_marcher.py:20-63):
midpoint-rule volume rendering of sorted samples.

ImportanceRenderer's GPU path does this inside the fused kernel; this module is the PyTorch form
used for CPU tensors, for autograd, and by callers that use the marcher on its own."""

import torch
import torch.nn as nn
import torch.nn.functional as F


class MipRayMarcher2(nn.Module):
    def __init__(self):
        super().__init__()

    def run_forward(self, colors, densities, depths, rendering_options):
        """colors [N,M,S,C], densities [N,M,S,1], depths [N,M,S,1] (ascending along S)
        -> composite_rgb [N,M,C] in (-1,1), composite_depth [N,M,1], weights [N,M,S-1,1]."""
        if rendering_options['clamp_mode'] != 'softplus':
            assert False, "MipRayMarcher only supports `clamp_mode`=`softplus`!"
        lo, hi = slice(None, -1), slice(1, None)
        span = depths[:, :, hi] - depths[:, :, lo]
        mid_color = (colors[:, :, lo] + colors[:, :, hi]) / 2
        mid_sigma = F.softplus((densities[:, :, lo] + densities[:, :, hi]) / 2 - 1)
        mid_depth = (depths[:, :, lo] + depths[:, :, hi]) / 2
        alpha = 1 - torch.exp(-(mid_sigma * span))
        survive = torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2)
        weights = alpha * torch.cumprod(survive, -2)[:, :, :-1]
        total = weights.sum(2)
        composite_rgb = torch.sum(weights * mid_color, -2)
        composite_depth = torch.sum(weights * mid_depth, -2) / total
        # zero-weight rays give NaN -> +inf -> the largest depth of the WHOLE call
        composite_depth = torch.nan_to_num(composite_depth, float('inf'))
        composite_depth = torch.clamp(composite_depth, torch.min(depths), torch.max(depths))
        if rendering_options.get('white_back', False):
            composite_rgb = composite_rgb + 1 - total
        composite_rgb = composite_rgb * 2 - 1
        return composite_rgb, composite_depth, weights

    def forward(self, colors, densities, depths, rendering_options):
        return self.run_forward(colors, densities, depths, rendering_options)
