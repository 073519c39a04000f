"""world_size-2 gloo tests of the one-process-per-GPU harness (frame sharding, the single gather, the timing
reduction) and the orbit cameras.  CPU only; the same code runs with backend nccl (= RCCL) on GPUs."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'g-nerf_amd'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import gnerf_harness as H
    r, w, _ = H.init_from_env()
    assert (r, w) == (rank, world)

    def render_frame(i, c2w):                       # stand-in renderer: frame content encodes the frame index and the camera
        img = torch.full((4, 6, 3), float(i % 256))
        img[0, 0, 0] = float(int(abs(c2w[0, 0, 3]) * 100) % 256)
        return img.to(torch.uint8)

    frames, (lo, hi) = H.render_orbit(render_frame, n_frames, rank, world)
    full = H.gather_frames(frames, n_frames, dst=0)
    t = H.max_over_ranks(1.0 + rank)
    q.put((rank, lo, hi, None if full is None else full.numpy(), t))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_frames', [8, 7])
def test_two_rank_orbit_gather(n_frames):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        rank, lo, hi, full, t = q.get(timeout=120)
        res[rank] = (lo, hi, full, t)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][0] == 0 and res[0][1] == res[1][0] and res[1][1] == n_frames          # contiguous cover
    assert abs((res[0][1] - res[0][0]) - (res[1][1] - res[1][0])) <= 1
    assert res[1][2] is None
    full = res[0][2]
    assert full.shape == (n_frames, 4, 6, 3) and full.dtype == np.uint8
    assert [int(f[1, 1, 1]) for f in full] == list(range(n_frames))                       # in order, each frame once
    assert res[0][3] == 2.0 and res[1][3] == 2.0                                          # max over ranks


def test_shard_range_covers_everything():
    import gnerf_harness as H
    for n in (0, 1, 7, 8, 240, 241):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = H.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
    assert H.shard_range(240, 3, 8) == (90, 120)            # BASELINE.json config 4: 30 consecutive frames per GPU


def test_orbit_cameras_match_reference(golden):
    import gnerf_harness as H
    g = golden('camera.npz')
    for i, ref in zip(g['orbit_frames'], g['orbit_cam2world']):
        np.testing.assert_allclose(H.orbit_pose(int(i), 120, 2.7)[0].numpy(), ref, atol=3e-7)
    c = H.camera_label(H.orbit_pose(0))
    assert c.shape == (1, 25) and abs(float(c[0, 16]) - 4.2647) < 1e-6 and float(c[0, 24]) == 1.0
    img = torch.tensor([[[[-1.0, 0.0], [1.0, 2.0]]]])
    assert H.to_uint8(img).flatten().tolist() == [0, 128, 255, 255]        # (x*127.5+128).clamp(0,255) truncated
