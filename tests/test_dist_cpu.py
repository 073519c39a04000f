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


# ---- data-parallel training step (BASELINE config 5): flat-gradient exchange, weight broadcast, consistency check ----

def _train_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'g-nerf_amd'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    import gnerf_harness as H
    import train_step_mi355x as T
    H.init_from_env()
    torch.manual_seed(10 + rank)                                      # deliberately DIFFERENT weights per rank ...
    opts = dict(T.RENDERING, depth_resolution=6, depth_resolution_importance=6)
    model = T.RendererTrainer(batch=1, plane_res=8, ballast_floats=100, rendering=opts)
    w_before = model.decoder.net[0].weight.detach().clone()
    H.broadcast_module(model)                                         # ... made equal to rank 0's
    H.check_ddp_consistency(model)
    w_rank0 = model.decoder.net[0].weight.detach().clone()
    c, target, target_depth = T.synthetic_batch(1, 4, torch.device('cpu'), seed=100 + rank)
    torch.manual_seed(1000 + rank)
    # local gradients, then the exchange by hand-made reference: mean over ranks, NaN/inf scrubbed
    img, depth = model(c, 4)
    ((img - target).abs().mean() + (depth - target_depth).abs().mean() + 1e-3 * model.ballast.sum()).backward()
    if rank == 1:
        model.ballast.grad[3] = float('nan')
        model.ballast.grad[4] = float('inf')
    params = H.params_with_grad(model)
    local = torch.cat([p.grad.flatten() for p in params]).clone()
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = torch.nan_to_num(torch.stack(gathered).sum(0) / world, nan=0, posinf=1e5, neginf=-1e5)
    flat = H.allreduce_flat_grads(params, bucket_bytes=4096)      # several async buckets
    ok_bucketed = bool(torch.allclose(flat, want, rtol=1e-6, atol=1e-9)) and all(p.grad.shape == p.shape for p in params)
    for p, piece in zip(params, local.split([p.numel() for p in params])):
        p.grad = piece.reshape(p.shape).clone()
    flat1 = H.allreduce_flat_grads(params)                                                 # the reference's single collective
    ok_single = bool(torch.allclose(flat1, want, rtol=1e-6, atol=1e-9))
    ofs = 0
    for p_ in params:
        if p_ is model.ballast:
            break
        ofs += p_.numel()
    scrubbed = (float(flat1[ofs + 3]), float(flat1[ofs + 4]))
    # two full optimiser steps keep the replicas identical although the data differ per rank
    opt = torch.optim.Adam(model.parameters(), lr=0.01, betas=(0.0, 0.99), eps=1e-8)
    for _ in range(2):
        loss = T.train_step(model, opt, c, target, target_depth, 4, bucket_bytes=4096)
    H.check_ddp_consistency(model)
    moved = float((model.decoder.net[0].weight.detach() - w_rank0).abs().max())
    # a diverged replica is caught
    caught = False
    if rank == 1:
        with torch.no_grad():
            model.decoder.net[2].bias[0] += 1.0
    try:
        H.check_ddp_consistency(model)
    except AssertionError as e:
        caught = 'bias' in str(e)
    q.put((rank, ok_bucketed, ok_single, scrubbed, moved, caught, float((w_before - w_rank0).abs().max()), float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=300)
        res[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        ok_bucketed, ok_single, scrubbed, moved, caught, init_diff, loss = res[r]
        assert ok_bucketed and ok_single
        assert scrubbed[0] == 0.0                                                        # NaN + x -> NaN -> 0
        assert scrubbed[1] == 1e5                                                        # inf -> 1e5
        assert moved > 0 and np.isfinite(loss)
    assert res[0][5] == 0.0 and res[1][5] > 0            # rank 1 started from different weights and now holds rank 0's
    assert res[1][4] and not res[0][4]                   # the diverged replica (rank 1) is the one that notices


def _run_bench(args, env_extra=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS='1', **(env_extra or {}))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + args, capture_output=True, text=True, env=env, cwd=root, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, [json.loads(ln) for ln in lines]


def test_bench_self_launch_eight_ranks_stub():
    """`python bench.py --gpus 8` WITHOUT a launcher (the form the driver records for N = 1): the parent starts its eight ranks itself,
    they rendezvous (gloo here, RCCL on GPUs), run the barriers / max-over-ranks / per-rank gather around a stub step, and exactly one
    JSON line comes back naming the backend, the ranks the group saw and each rank's device."""
    r, lines = _run_bench(['--gpus', '8', '--steps', '4', '--warmup', '1', '--stub-step'])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout
    line = lines[0]
    assert line['stub'] is True and line['value'] is None                      # a rehearsal line can not be mistaken for a measurement
    assert line['n_gpus'] == 8 and line['ranks_seen'] == 8 and line['backend'] == 'gloo' and line['self_launched'] is True
    assert [x['rank'] for x in line['ranks']] == list(range(8)) and len({x['pid'] for x in line['ranks']}) == 8


def test_one_solver_search_per_node_eight_ranks_stub(tmp_path):
    """Round 6 (bench.one_solver_search): rank 0 runs the warm-up -- MIOpen's solver search -- alone, its user database / kernel cache are
    copied into every other rank's own directories, and only then do the other ranks issue their first convolution.  Rehearsed with the
    stub step: the stub "convolution" searches for GNERF_BENCH_STUB_SEARCH_S seconds unless its rank's MIOPEN_USER_DB_PATH already holds
    the solver record.  With eight ranks exactly one search is paid (rank 0's), every other rank finds the record BEFORE its first
    convolution, and the wall time of the phase does not grow with the number of ranks."""
    env = {'TMPDIR': str(tmp_path), 'GNERF_BENCH_STUB_SEARCH_S': '1.5'}
    walls = {}
    for n in (2, 8):
        import shutil
        shutil.rmtree(os.path.join(str(tmp_path), f'gnerf_miopen_{os.getuid()}'), ignore_errors=True)
        r, lines = _run_bench(['--gpus', str(n), '--steps', '2', '--warmup', '1', '--stub-step'], env)
        assert r.returncode == 0 and len(lines) == 1, r.stderr[-3000:]
        ranks = lines[0]['ranks']
        assert [x['rank'] for x in ranks] == list(range(n))
        assert ranks[0]['solver_search']['hit_before_first_conv'] is False and ranks[0]['solver_search']['searched_s'] == 1.5
        for x in ranks[1:]:
            assert x['solver_search']['hit_before_first_conv'] is True and x['solver_search']['searched_s'] == 0.0, x
        walls[n] = max(x['solver_search']['wall_s'] for x in ranks)
    assert walls[8] < walls[2] + 1.0 and walls[8] < 2 * 1.5, walls            # one search, not eight (eight in a row would be 12 s)


def test_one_solver_search_leaves_hand_set_directories_alone(tmp_path):
    """A MIOPEN_USER_DB_PATH the caller chose (not per_rank_miopen_env's rank<k> layout) is not copied anywhere: a shared directory needs none."""
    import bench
    mine = tmp_path / 'shared_db'
    mine.mkdir()
    (mine / 'x.ufdb.txt').write_text('record')
    calls = []
    out = bench.one_solver_search(0, 3, lambda: calls.append('warm') or 7, lambda: calls.append('barrier'), env={'MIOPEN_USER_DB_PATH': str(mine)})
    assert out == 7 and calls == ['warm', 'barrier'] and sorted(os.listdir(tmp_path)) == ['shared_db']
    # the per-rank layout IS copied, before the barrier releases the others
    base = tmp_path / 'gnerf_miopen_0'
    for k in range(3):
        (base / f'rank{k}' / 'db').mkdir(parents=True)
    (base / 'rank0' / 'db' / 'gfx950.ufdb.txt').write_text('searched')
    order = []
    bench.one_solver_search(0, 3, lambda: order.append('warm'), lambda: order.append(sorted(os.listdir(base / 'rank2' / 'db'))),
                            env={'MIOPEN_USER_DB_PATH': str(base / 'rank0' / 'db')})
    assert order == ['warm', ['gfx950.ufdb.txt']]
    # a rank other than 0 waits first, then warms
    order = []
    bench.one_solver_search(2, 3, lambda: order.append('warm'), lambda: order.append('barrier'), env={})
    assert order == ['barrier', 'warm']


def test_bench_self_launch_reports_a_dead_rank():
    """A rank that dies leaves its peers in a collective: the launcher must end them and exit non-zero, with no result line."""
    r, lines = _run_bench(['--gpus', '3', '--steps', '2', '--warmup', '1', '--stub-step'],
                          {'GNERF_BENCH_STUB_FAIL_RANK': '1', 'GNERF_BENCH_PEER_GRACE_S': '3'}, timeout=300)
    assert r.returncode != 0 and not lines, (r.returncode, r.stdout)
    assert 'rank exit codes' in r.stderr


def test_bench_under_outer_launcher_is_not_relaunched():
    """Under torch.distributed.run (WORLD_SIZE set) bench.py is a plain rank: no second generation of processes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--stub-step'],
                       capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1 and lines[0]['ranks_seen'] == 2 and lines[0]['self_launched'] is False


def _fake_kfd(tmp_path, nodes):
    """A driver topology tree like /sys/class/kfd/kfd/topology/nodes: nodes = [(simd_count, drm_render_minor, render node present)]."""
    base, dri = tmp_path / 'nodes', tmp_path / 'dri'
    dri.mkdir()
    for i, (simd, minor, present) in enumerate(nodes):
        d = base / str(i)
        d.mkdir(parents=True)
        (d / 'properties').write_text(f'cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\ndrm_render_minor {minor}\n')
        if present and minor:
            (dri / f'renderD{minor}').write_text('')
    return str(base), str(dri)


def test_visible_gpus_are_counted_from_the_driver_topology(tmp_path, monkeypatch):
    """The self-launching parent counts GPUs from the kernel driver's files: CPU nodes (simd_count 0) and GPUs whose render node this
    container was not given do not count, and the *_VISIBLE_DEVICES lists cap the number."""
    import bench
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(var, raising=False)
    sysfs, dri = _fake_kfd(tmp_path, [(0, 0, False), (0, 0, False)] + [(1024, 128 + i, i != 5) for i in range(8)])
    assert bench.visible_gpus_no_hip(sysfs, dri) == 7
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,1,2')
    assert bench.visible_gpus_no_hip(sysfs, dri) == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '')
    assert bench.visible_gpus_no_hip(sysfs, dri) == 0
    assert bench.visible_gpus_no_hip(str(tmp_path / 'absent'), dri) is None            # unreadable topology: nothing is refused


def test_self_launching_parent_never_asks_torch_for_the_device_count(monkeypatch):
    """`python bench.py --gpus N` without a launcher: the parent decides and spawns without one GPU-runtime call -- with
    torch.cuda.device_count (whose amdsmi route falls through to hipGetDeviceCount when amdsmi fails) and the HIP initialisation
    rigged to raise, it still reaches self_launch, and refuses readably when the driver's topology shows fewer GPUs than ranks."""
    import sys
    import bench

    def boom(*a, **k):
        raise AssertionError('the self-launching parent touched the GPU runtime')
    monkeypatch.setattr(torch.cuda, 'device_count', boom)
    monkeypatch.setattr(torch.cuda, 'is_available', boom)
    monkeypatch.setattr(torch.cuda, '_lazy_init', boom)
    if hasattr(torch._C, '_cuda_getDeviceCount'):
        monkeypatch.setattr(torch._C, '_cuda_getDeviceCount', boom)
    for var in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'GNERF_DIST_BACKEND'):
        monkeypatch.delenv(var, raising=False)
    launched = []
    monkeypatch.setattr(bench, 'self_launch', lambda n, argv: launched.append((n, list(argv))) or 0)
    monkeypatch.setattr(bench, 'visible_gpus_no_hip', lambda *a: 8)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '2'])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and launched == [(8, ['--gpus', '8', '--steps', '2'])]
    monkeypatch.setattr(bench, 'visible_gpus_no_hip', lambda *a: 1)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert '1 GPU(s) visible' in str(e.value.code) and len(launched) == 1


def test_every_rank_gets_its_own_miopen_directories(tmp_path, monkeypatch):
    """Ranks that search MIOpen solvers at the same time must not share one user database / kernel cache."""
    import bench
    monkeypatch.setattr('tempfile.tempdir', str(tmp_path))
    envs = [bench.per_rank_miopen_env(r, {}) for r in range(3)]
    assert len({e['MIOPEN_USER_DB_PATH'] for e in envs}) == 3 and len({e['MIOPEN_CUSTOM_CACHE_DIR'] for e in envs}) == 3
    assert all(os.path.isdir(e['MIOPEN_USER_DB_PATH']) for e in envs)
    kept = bench.per_rank_miopen_env(0, {'MIOPEN_USER_DB_PATH': '/somewhere'})
    assert kept['MIOPEN_USER_DB_PATH'] == '/somewhere'                                  # a caller's choice is left alone


def test_cpu_topology_reads_this_host():
    import bench
    t = bench.host_cpu_topology()
    assert t['logical_cpus'] >= 1 and (t['physical_cores'] is None or 1 <= t['physical_cores'] <= t['logical_cpus'])
