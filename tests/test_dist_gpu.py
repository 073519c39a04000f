"""RCCL variants of the 2-rank harness tests (tests/test_dist_cpu.py runs the same code over gloo): one process per GPU,
backend nccl (= RCCL over xGMI).  They run by themselves wherever at least two GPUs are visible and are skipped on a
one-GPU box (the driver's scaling run on a whole node is the other place RCCL sees N > 1 ranks)."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _n_gpus():
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_n_gpus() < 2, reason='needs two GPUs (RCCL with more than one rank)')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, 'g-nerf_amd'), root]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    os.environ.pop('GNERF_DIST_BACKEND', None)
    import torch.distributed as dist
    import gnerf_harness as H
    import gnerf_hip
    import train_step_mi355x as T
    r, w, local = H.init_from_env()                                     # nccl when a GPU is visible
    assert dist.get_backend() == 'nccl'
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    # (1) frames sharded over the ranks, one uint8 gather
    frames = torch.full((3, 8, 8, 3), rank, dtype=torch.uint8, device=dev)
    full = H.gather_frames(frames, 6, dst=0)
    ok_gather = (rank != 0) or (full.shape == (6, 8, 8, 3) and full[:3].eq(0).all().item() and full[3:].eq(1).all().item())
    # (2) the render kernels on the SECOND device of this node (per-device LDS attribute, workspace, stream handling)
    torch.manual_seed(5)
    planes = torch.randn(1, 3, 32, 16, 16, device=dev)
    model = T.RendererTrainer(batch=1, plane_res=16, ballast_floats=64, rendering=dict(T.RENDERING, depth_resolution=130, depth_resolution_importance=100)).to(dev)
    H.broadcast_module(model)
    H.check_ddp_consistency(model)
    c, target, target_depth = T.synthetic_batch(1, 4, dev, seed=100 + rank)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, betas=(0.0, 0.99))
    for _ in range(2):                                                  # generic kernel (>64 KB of LDS) + backward kernel + RCCL all-reduce
        loss = T.train_step(model, opt, c, target, target_depth, 4, bucket_bytes=4096)
    H.check_ddp_consistency(model)
    t = H.max_over_ranks(1.0 + rank, dev)
    q.put((rank, bool(ok_gather), float(loss), t, gnerf_hip.last_mlp_choice(dev)))
    dist.barrier()
    dist.destroy_process_group()


@needs_two
def test_two_rank_rccl_gather_and_training_step():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok_gather, loss, t, _ in res:
        assert ok_gather and loss == loss and t == 2.0
