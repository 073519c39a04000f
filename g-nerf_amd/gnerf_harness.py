"""Host-side pieces of the inference harness that surround the hot path: the camera
orbit of gen_videos.py, uint8 frame conversion, and one-process-per-GPU sharding of frames/rays with the single
collective the path needs (a gather of uint8 frames).  No device kernels here; the renderer itself is
training.volumetric_rendering.renderer.ImportanceRenderer.

Reference behaviour reproduced (g_nerf/):
  * camera for frame i of n: LookAtPoseSampler.sample(3.14/2 + 0.7 sin(2*3.14*i/n), 3.14/2 - 0.05 + 0.3 cos(2*3.14*i/n),
    radius)  -- gen_videos.py:155-158 (note 3.14, not pi), camera_utils.py:89-106,155-174
  * intrinsics [[4.2647,0,.5],[0,4.2647,.5],[0,0,1]] -- gen_videos.py:135
  * c = cat(cam2world.reshape(16), intrinsics.reshape(9)) -- gen_videos.py:170
  * uint8 = (img * 127.5 + 128).clamp(0, 255) -- gen_videos.py:173
"""

import math
import os

import torch

FFHQ_INTRINSICS = ((4.2647, 0.0, 0.5), (0.0, 4.2647, 0.5), (0.0, 0.0, 1.0))


def lookat_pose(yaw, pitch, radius, device='cpu'):
    """cam2world [1,4,4] of a camera on a sphere of `radius` looking at the origin, y up, no roll."""
    dt = torch.float32
    theta = torch.tensor([[yaw]], dtype=dt)
    phi = torch.tensor([[pitch]], dtype=dt)
    org = torch.zeros(1, 3, dtype=dt)
    org[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - theta)
    org[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - theta)
    org[:, 1:2] = radius * torch.cos(phi)

    def unit(v):
        return v / torch.norm(v, dim=-1, keepdim=True)

    fwd = unit(unit(-org))
    up = torch.tensor([[0.0, 1.0, 0.0]], dtype=dt)
    right = -unit(torch.cross(up, fwd, dim=-1))
    up2 = unit(torch.cross(fwd, right, dim=-1))
    rot = torch.eye(4, dtype=dt)[None].clone()
    rot[:, :3, :3] = torch.stack((right, up2, fwd), dim=-1)
    trans = torch.eye(4, dtype=dt)[None].clone()
    trans[:, :3, 3] = org
    return (trans @ rot).to(device)


def orbit_pose(i, frame_num=120, radius=2.7, yaw_range=0.7, pitch_range=0.3, device='cpu'):
    """cam2world of frame i of gen_videos.py's orbit."""
    yaw = 3.14 / 2 + yaw_range * math.sin(2 * 3.14 * i / frame_num)
    pitch = 3.14 / 2 - 0.05 + pitch_range * math.cos(2 * 3.14 * i / frame_num)
    return lookat_pose(yaw, pitch, radius, device)


def orbit_labels(frames, frame_num=120, radius=2.7, yaw_range=0.7, pitch_range=0.3, intrinsics=FFHQ_INTRINSICS):
    """camera_label(orbit_pose(i, ...)) for every i of `frames`, [len(frames), 25] on the host, in ONE set of batched tensor ops: the
    per-frame form costs ~0.15 ms of host time per camera (a 240-frame orbit: 30-45 ms in front of a 130 ms orbit).  Same arithmetic
    per element as lookat_pose, so the rows equal the per-frame labels bit for bit (tests/test_host_cpu.py)."""
    dt = torch.float32
    frames = list(frames)
    theta = torch.tensor([[3.14 / 2 + yaw_range * math.sin(2 * 3.14 * i / frame_num)] for i in frames], dtype=dt)
    phi = torch.tensor([[3.14 / 2 - 0.05 + pitch_range * math.cos(2 * 3.14 * i / frame_num)] for i in frames], dtype=dt)
    n = len(frames)
    org = torch.zeros(n, 3, dtype=dt)
    org[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - theta)
    org[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - theta)
    org[:, 1:2] = radius * torch.cos(phi)

    def unit(v):
        return v / torch.norm(v, dim=-1, keepdim=True)

    fwd = unit(unit(-org))
    up = torch.tensor([[0.0, 1.0, 0.0]], dtype=dt).expand(n, -1)
    right = -unit(torch.cross(up, fwd, dim=-1))
    up2 = unit(torch.cross(fwd, right, dim=-1))
    rot = torch.eye(4, dtype=dt)[None].repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((right, up2, fwd), dim=-1)
    trans = torch.eye(4, dtype=dt)[None].repeat(n, 1, 1)
    trans[:, :3, 3] = org
    return camera_label(trans @ rot, intrinsics)


def camera_label(cam2world, intrinsics=FFHQ_INTRINSICS):
    """The 25-float conditioning vector c the generator's synthesis() takes."""
    k = torch.tensor(intrinsics, dtype=torch.float32, device=cam2world.device)
    n = cam2world.shape[0]
    return torch.cat([cam2world.reshape(n, 16), k.reshape(1, 9).expand(n, -1)], 1)


def to_uint8(img):
    """[-1,1] float image(s) [N,C,H,W] -> uint8 [N,H,W,C] like gen_videos.py:173."""
    if img.is_cuda and img.dtype == torch.float32 and img.ndim == 4 and img.shape[1] <= 64 and not (torch.is_grad_enabled() and img.requires_grad):
        import gnerf_hip
        return gnerf_hip.to_uint8_nhwc(img)             # the same arithmetic in one launch (csrc/planes.hip)
    return (img * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()


# ---------------------------------------------------------------------------------------------
# one process per GPU


def configure_backend(solver_search=None):
    """Process-wide library settings of the harnesses (gen_videos_mi355x.py, train_step_mi355x.py, bench.py's secondary, tools/).
    solver_search: torch.backends.cudnn.benchmark -- on ROCm, MIOpen times its applicable solvers the first time it sees a
    convolution shape and keeps the fastest, instead of taking its heuristic's first pick.  The reference's training loop turns it
    on (training_loop.py:133,144); gen_videos.py leaves torch's default.  The generator's convolutions are 60 % of an orbit frame's
    GPU time and the searched solvers are that much better on gfx950 (orbit +16 %, config 3 +12 %: profiles/r03_generator.jsonl),
    at the price of a few seconds in the warm-up frame.  Default: on; GNERF_MIOPEN_FIND=0 turns it off."""
    if solver_search is None:
        solver_search = os.environ.get('GNERF_MIOPEN_FIND', '1') != '0'
    torch.backends.cudnn.benchmark = bool(solver_search)
    return bool(solver_search)


def init_from_env():
    """(rank, world_size, local_rank).  Initialises torch.distributed from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*
    when WORLD_SIZE > 1: backend 'nccl' (= RCCL on ROCm) if a GPU is visible, else 'gloo'."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            # GNERF_DIST_BACKEND=gloo forces gloo although a GPU is visible (to exercise the multi-rank code on a one-GPU box)
            use_gpu = os.environ.get('GNERF_DIST_BACKEND', 'nccl') == 'nccl' and torch.cuda.is_available()
            if use_gpu:
                torch.cuda.set_device(local_rank)
                dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
            else:
                dist.init_process_group('gloo')
    return rank, world, local_rank


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of n_items owned by `rank`: sizes differ by at most one, earlier ranks get the extras
    (240 frames over 8 ranks -> 30 consecutive frames each, BASELINE.json config 4)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_frames(frames, n_total, dst=0):
    """frames: this rank's uint8 block [n_local, ...] (its shard_range of n_total).  Returns the full [n_total, ...] tensor on
    rank `dst` (None elsewhere).  ONE collective, a `gather` to `dst` of equal-size padded uint8 blocks (RCCL over xGMI on GPUs):
    every other rank sends its block once over its own link to `dst` -- 189 MB arrive there for 240 frames of 512x512x3, 23.6 MB
    per link at 8 GPUs; an all_gather would deliver those 189 MB to every rank, 8x the traffic, for nothing."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return frames
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    pad = torch.zeros((per,) + tuple(frames.shape[1:]), dtype=frames.dtype, device=frames.device)
    pad[:frames.shape[0]] = frames
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, gather_list=out, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r][:hi - lo])
    return torch.cat(parts, 0)


def max_over_ranks(value, device='cpu'):
    """max of a python float over all ranks (the elapsed-time reduction of bench.py)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if dist.get_backend() != 'nccl':
        device = 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def render_orbit(render_frame, n_frames, rank=0, world=1, **orbit_kwargs):
    """Render this rank's block of an n_frames orbit.  render_frame(i, cam2world) -> uint8 [H,W,3] (or [1,H,W,3]).
    Returns (frames uint8 [n_local,H,W,3], (lo, hi))."""
    lo, hi = shard_range(n_frames, rank, world)
    frames = []
    for i in range(lo, hi):
        f = render_frame(i, orbit_pose(i, frame_num=n_frames, **orbit_kwargs))
        frames.append(f.reshape((1,) + tuple(f.shape[-3:])))
    if not frames:
        return None, (lo, hi)
    return torch.cat(frames, 0), (lo, hi)


# ---------------------------------------------------------------------------------------------
# data-parallel training step (BASELINE config 5): the reference's collective call sites, re-done for xGMI


def _dist_on():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def params_with_grad(*modules_or_tensors):
    """Parameters that take part in the gradient exchange, in a fixed order (training_loop.py:380-386: numel > 0 and a
    gradient present)."""
    out = []
    for m in modules_or_tensors:
        ps = m.parameters() if isinstance(m, torch.nn.Module) else [m]
        out += [p for p in ps if p.numel() > 0 and p.grad is not None]
    return out


def allreduce_flat_grads(params, bucket_bytes=None):
    """Average the gradients of `params` over all ranks and scrub non-finite values, with the semantics of the
    reference's manual exchange (training_loop.py:388-396, :427-436): flatten -> SUM all-reduce -> / world ->
    nan_to_num(nan=0, posinf=1e5, neginf=-1e5) -> hand each parameter a view of the flat vector as its .grad.

    What is different from the reference: the flat vector is exchanged in `bucket_bytes` pieces issued back to back as
    asynchronous collectives (RCCL pipelines them over the 7 xGMI links; the reference issues one monolithic blocking
    all-reduce of ~123 MB), and the division + scrub of a bucket runs while the next bucket is on the wire.
    bucket_bytes=None -> one collective (the reference's pattern).  Returns the flat vector."""
    import torch.distributed as dist
    params = list(params)
    if not params:
        return None
    flat = torch.cat([p.grad.flatten() for p in params])
    if _dist_on():
        world = dist.get_world_size()
        if bucket_bytes is None or flat.numel() * flat.element_size() <= bucket_bytes:
            dist.all_reduce(flat)
            flat /= world
            torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
        else:
            per = max(1, bucket_bytes // flat.element_size())
            pieces = list(flat.split(per))
            works = [dist.all_reduce(pc, async_op=True) for pc in pieces]
            for pc, w in zip(pieces, works):
                w.wait()
                pc /= world
                torch.nan_to_num(pc, nan=0, posinf=1e5, neginf=-1e5, out=pc)
    else:
        torch.nan_to_num(flat, nan=0, posinf=1e5, neginf=-1e5, out=flat)
    for p, g in zip(params, flat.split([p.numel() for p in params])):
        p.grad = g.reshape(p.shape)
    return flat


def broadcast_module(module, src=0):
    """Make every parameter and buffer of `module` equal to rank `src`'s (training_loop.py:234-238).  The reference
    broadcasts 195 tensors one by one; here tensors of one dtype travel as ONE flat buffer."""
    import torch.distributed as dist
    if not _dist_on():
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.numel() > 0]
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dt, ts in by_dtype.items():
        flat = torch.cat([t.detach().flatten() for t in ts])
        dist.broadcast(flat, src=src)
        for t, piece in zip(ts, flat.split([t.numel() for t in ts])):
            t.detach().copy_(piece.reshape(t.shape))


def check_ddp_consistency(module, src=0):
    """Assert that every parameter / buffer equals rank `src`'s bit for bit (misc.py:202-213, run at snapshot time,
    training_loop.py:520-521) -- one flat broadcast per dtype instead of one per tensor."""
    import torch.distributed as dist
    if not _dist_on():
        return
    named = [(n, t) for n, t in list(module.named_parameters()) + list(module.named_buffers()) if t.numel() > 0]
    by_dtype = {}
    for n, t in named:
        by_dtype.setdefault(t.dtype, []).append((n, t))
    for dt, nts in by_dtype.items():
        mine = torch.cat([(torch.nan_to_num(t.detach()) if t.is_floating_point() else t.detach()).flatten() for _, t in nts])
        other = mine.clone()
        dist.broadcast(other, src=src)
        if not torch.equal(mine, other):
            ofs = 0
            for n, t in nts:
                if not torch.equal(mine[ofs:ofs + t.numel()], other[ofs:ofs + t.numel()]):
                    raise AssertionError(f'{type(module).__name__}.{n} differs from rank {src}')
                ofs += t.numel()


class FullyConnected(torch.nn.Module):
    """The layer OSGDecoder is made of (networks_stylegan2.py:101-134, 'linear' activation with bias): weight ~ N(0,1) /
    lr_multiplier, runtime gains weight_gain = lr_multiplier / sqrt(in), bias_gain = lr_multiplier."""

    def __init__(self, in_features, out_features, lr_multiplier=1.0):
        super().__init__()
        self.in_features, self.out_features, self.activation = in_features, out_features, 'linear'
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.zeros([out_features]))
        self.weight_gain = lr_multiplier / (in_features ** 0.5)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        return torch.addmm((self.bias * self.bias_gain).unsqueeze(0), x, (self.weight * self.weight_gain).t())


class TriPlaneDecoder(torch.nn.Module):
    """OSGDecoder (triplane.py:113-136) for harnesses that must run where the reference tree is absent (the GPU box):
    mean over the three planes -> FC(32,64) -> Softplus -> FC(64, 1+32); sigma = out[0], rgb = sigmoid(out[1:])*1.002-0.001."""

    def __init__(self, n_features=32, decoder_lr_mul=1.0, decoder_output_dim=32):
        super().__init__()
        self.net = torch.nn.Sequential(FullyConnected(n_features, 64, decoder_lr_mul), torch.nn.Softplus(),
                                       FullyConnected(64, 1 + decoder_output_dim, decoder_lr_mul))

    def forward(self, sampled_features, ray_directions):
        x = sampled_features.mean(1)
        N, M, C = x.shape
        x = self.net(x.view(N * M, C)).view(N, M, -1)
        return {'rgb': torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001, 'sigma': x[..., 0:1]}
