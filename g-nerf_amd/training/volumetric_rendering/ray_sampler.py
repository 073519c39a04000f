"""RaySampler with the reference's interface (training/volumetric_rendering/ray_sampler.py:18-63):
camera matrices (OpenCV convention) -> one ray through each pixel centre.

GPU tensors go to the hand-written kernel (gnerf_make_rays); CPU tensors are computed with
PyTorch ops, as the reference does everywhere."""

import torch

import gnerf_hip


class RaySampler(torch.nn.Module):
    def __init__(self):
        super().__init__()
        # attributes the reference creates (ray_sampler.py:21); kept so that pickles round-trip
        self.ray_origins_h, self.ray_directions, self.depths, self.image_coords, self.rendering_options = None, None, None, None, None

    def forward(self, cam2world_matrix, intrinsics, resolution):
        """cam2world_matrix [N,4,4], intrinsics [N,3,3], resolution int
        -> ray_origins [N,res*res,3], ray_dirs [N,res*res,3]; ray m = row*res + col."""
        if cam2world_matrix.device.type == 'cuda' and not (torch.is_grad_enabled() and (cam2world_matrix.requires_grad or intrinsics.requires_grad)):
            return gnerf_hip.make_rays(cam2world_matrix, intrinsics, resolution)
        return _make_rays_torch(cam2world_matrix, intrinsics, resolution)


def _make_rays_torch(cam2world, intrinsics, res):
    n = cam2world.shape[0]
    dev = cam2world.device
    fx, fy = intrinsics[:, 0, 0, None], intrinsics[:, 1, 1, None]
    cx, cy = intrinsics[:, 0, 2, None], intrinsics[:, 1, 2, None]
    sk = intrinsics[:, 0, 1, None]
    ticks = torch.arange(res, dtype=torch.float32, device=dev) * (1. / res) + (0.5 / res)
    x_cam = ticks.repeat(res)[None].expand(n, -1)                  # column index runs fastest
    y_cam = ticks.repeat_interleave(res)[None].expand(n, -1)
    x_lift = (x_cam - cx + cy * sk / fy - sk * y_cam / fy) / fx
    y_lift = (y_cam - cy) / fy
    ones = torch.ones_like(x_lift)
    pts = torch.stack((x_lift, y_lift, ones, ones), dim=-1)        # homogeneous points at z = 1
    world = torch.bmm(cam2world, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    cam_loc = cam2world[:, :3, 3]
    dirs = torch.nn.functional.normalize(world - cam_loc[:, None, :], dim=2)
    origins = cam_loc.unsqueeze(1).repeat(1, dirs.shape[1], 1)
    return origins, dirs
