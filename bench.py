#!/usr/bin/env python3
"""Headline benchmark: rays/sec of the tri-plane importance renderer on BASELINE.json's config 2
(renderer-only: 128x128 rays x (48+48) samples, 32-channel 256x256 tri-planes, batch 4 per GPU).

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over one batch, what ImportanceRenderer.forward does per call on the
GPU when it is handed BASELINE.md section 2's input -- planes `torch.randn(4,3,32,256,256)`, i.e. the
reference API's NCHW tensor: ray generation from the cameras, the NCHW -> [3N,H,W,32] layout change with
max |planes| riding on it, the reference's two uniform draws (torch.rand, same shapes/order as
renderer.py:190/:241), the device-side choice of the decoder arithmetic, and the fused render kernel (+ its
depth-clamp epilogue).  Inputs (planes, decoder, cameras) are resident in HBM before the timed region.
`value` is that step.  `producer_layout_step` (top level, and `config.producer_layout_value`) is the same
step when the planes arrive in the layout this repo's plane producer writes (channels_last [N,H,W,96] +
max |planes|, read in place: no layout change inside the step; `--producer-layout` makes it the headline).
Rays shard embarrassingly: each rank renders its own batch (weak scaling, no data-path collective); the only
collectives are the barriers bracketing the timed region and the max-reduce of the elapsed time.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = render_kernel, timed with HIP events
on the launch stream inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample
of the same workload, on this host's cores).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]

import torch  # noqa: E402

# config 2 (BASELINE.json / SURVEY.md section 8d)
N_ITEMS, RES, S_COARSE, S_FINE, PLANE = 4, 128, 48, 48, 256
# GNERF_BENCH_FUSED_PREP=0: the step's rays and draws as three launches (gnerf_make_rays + two torch.rand), as until round 5
FUSED_PREP = os.environ.get('GNERF_BENCH_FUSED_PREP', '1') != '0'
RAY_START, RAY_END, BOX_WARP = 2.25, 3.3, 1.0

# Algorithmic work per ray (DESIGN.md section 4 "Roofline accounting")
FLOP_MLP_PER_SAMPLE = 2 * 32 * 64 + 2 * 64 * 33           # 8320, the MFMA-eligible contraction
GATHER_BYTES_PER_SAMPLE = 12 * 32 * 4                     # 12 bilinear taps x 32 fp32 channels (cache-level traffic)
PEAK_FP32_MFMA_TFLOPS = 157.3                             # MI355X_MICROARCH.md, v_mfma_f32_16x16x4_f32
PEAK_F16_MFMA_TFLOPS = 2500.0                             # dense f16/bf16 matrix peak
PEAK_HBM_GBS = 8000.0
PEAK_L2_GBS = 34500.0                                     # aggregate L2 read bandwidth (MI355X_MICROARCH.md)
N_SIMD, CLOCK_HZ = 1024, 2.4e9
# SIMD cycles per wave64 instruction at the render kernel's occupancy (4 waves per SIMD), MEASURED on the device by
# tools/probes/valu_issue_probe.hip (profiles/r03_valu_issue_probe.json; s_memtime around streams of independent instructions).
# Four classes -- the guide's "2 cycles (SIMD-32) / 4 for one wave alone" turned out to depend on the OPCODE, not on the wave count:
#   full     v_add / v_sub / v_mul / v_fma / v_fmac_f32, v_mov, v_and / v_or / v_xor, v_add / v_sub_u32, right shifts       2.3 - 2.6
#   half     v_min / v_max / v_med3, v_cmp, v_cndmask, conversions, v_floor, left shifts, 24-bit and 32-bit integer multiplies,
#            DPP moves / adds, v_readlane, v_fma_mix_f32, and every packed-fp32 op (v_pk_fma_f32 = two FMAs in 4.4)              4.2 - 4.7
#   quarter  v_exp / v_log / v_rcp_f32, v_fma_mixlo / hi_f16, v_permlane*_swap                                               8.3 - 8.9
#   mfma     v_mfma_f32_16x16x32_f16 16.5 (v_mfma_f32_16x16x4_f32 32.1); next to VALU work of other waves the costs ADD
#            (one MFMA + 8 v_fma_f32 per wave: 34.7 cycles against 16.5 + 8 x 2.4 = 36): the matrix pipe is not a free second port
# The fall-back numbers below are that file's values; roofline() re-reads the file when it is there.
ISSUE_CYCLES = {'full': 2.44, 'half': 4.40, 'quarter': 8.60, 'mfma_f16': 16.5}
PROBE_JSON = os.path.join(ROOT, 'profiles', 'r04_valu_issue_probe.json')       # (round 3's file plus the K = 16 f16 MFMA and v_mul_hi_u32 streams)
GATHER_BYTES_PER_CYCLE_PER_CU = 64.0                      # same probe: 8 lanes per 128-byte line, any line order, L1 or L2 resident


def issue_cycles():
    """(prices, source): SIMD cycles per wave64 instruction of each class at 4 waves per SIMD, from the committed probe."""
    try:
        rows = {(r['stream'], r['waves_per_simd']): r['simd_cyc_per_inst'] for r in json.load(open(PROBE_JSON))['streams']}
        mean = lambda names: sum(rows[(n, 4)] for n in names) / len(names)
        return ({'full': mean(['v_fma_f32', 'v_add_f32', 'v_mul_f32', 'v_mov_b32', 'v_add_u32', 'v_and_b32']),
                 'half': mean(['v_max_f32', 'v_cmp_lt_f32_vcc', 'v_cndmask_b32_sgpr', 'v_cvt_pk_f16_f32', 'v_fma_mix_f32', 'v_mad_u32_u24', 'v_floor_f32']),
                 'quarter': mean(['v_exp_f32', 'v_log_f32', 'v_rcp_f32']), 'mfma_f16': rows[('mfma_f16_16x16x32', 4)]},
                'profiles/' + os.path.basename(PROBE_JSON))
    except Exception:
        return dict(ISSUE_CYCLES), 'bench.py constants (probe file missing)'


def algorithmic_valu_per_ray(s=48, f=48):
    """Vector instructions a PERFECT schedule of this algorithm needs per ray: lane-operations per issue class (every lane useful,
    no address / staging / LDS hand-off overhead, nothing recomputed) and MFMA wave-instructions.  Numerator of roofline.frac."""
    n = s + f
    per_sample = {
        'full': 6                   # position o + t d (3 fma) and the box scale (3 mul)
                + 3 * 16            # per plane: pixel coordinates 2, fractions and 1 - f 4, four tap weights 4, x1 / y1 2, tap addresses 4
                + 12 * 32           # the blend: 12 taps x 32 channels, one FMA each (packed or not: 2.2 cycles per FMA either way)
                + 64                # softplus log2(1 + 2^p'): the add (round 4: the kernel runs this short form whenever the decoder's norms bound |p'|;
                                    # until then the table charged the overflow-safe form's two adds and one max, which the kernel executed)
                + 64                # density row: 64 FMAs
                + 32 * 2            # sigmoid 1.002 / (1 + 2^o) - 0.001: add, fma
                + 32,               # composite: one FMA per channel
        'half': 3 * 16              # per plane: floor 2, float->int 2, zero-padding compares + selects 8, clamps 4
                + 32 * 2            # hi/lo f16 split of the 32 features: 2 instructions per value (cvt_pk 1/2, fma_mix 1, cvt_pk 1/2)
                + 64 * 2,           # hi/lo split of the 64 activations
        'quarter': 64 * 2 + 32 * 2,     # exp2 + log2 per hidden unit, exp2 + rcp per colour
    }
    intervals = (s - 1) + (n - 1)                       # coarse march + final march (ray_marcher.py:26-42)
    per_ray = {'full': intervals * 20 + s * 4 + 64,     # marches (differences, midpoints, products), proposals, outputs
               'half': intervals * 10 + (f * 24 + 300) + (n * 12 + f * 10),      # scans (DPP), importance (compare + count), merge ranks
               'quarter': intervals * 3}                # softplus (exp2, log2) and exp of every interval
    out = {k: per_sample[k] * n + per_ray[k] for k in per_sample}
    out['mfma_f16'] = 24 * ((s + 15) // 16 + (f + 15) // 16)       # 24 v_mfma_f32_16x16x32_f16 per 16-sample tile (wave instructions)
    return out


def roofline(kernel_ms, rays, s, f, plane, n_items, pmc=None, pmc_source=None, traffic=None, kernel_name='render_kernel_pipe<1, auto, full>'):
    """The `roofline` object of the result line for the render kernel at `kernel_ms` per launch.  `pmc`: mean counters per
    launch from a committed rocprofv3 pass of this same command (profiles/rNN_final*_pmc.json; tools/prof_forward.sh)."""
    k_s = kernel_ms * 1e-3
    samples = rays * (s + f)
    flops = samples * FLOP_MLP_PER_SAMPLE
    alg = algorithmic_valu_per_ray(s, f)
    price, price_src = issue_cycles()
    alg_cycles_per_ray = sum(alg[k] / 64 * price[k] for k in ('full', 'half', 'quarter')) + alg['mfma_f16'] * price['mfma_f16']
    alg_floor_ms = alg_cycles_per_ray * rays / (N_SIMD * CLOCK_HZ) * 1e3
    gather_floor_ms = samples * GATHER_BYTES_PER_SAMPLE / (GATHER_BYTES_PER_CYCLE_PER_CU * 256 * CLOCK_HZ) * 1e3
    out = {
        'kernel': kernel_name, 'bound': 'valu_issue',
        # achieved = SIMD issue cycles of the ALGORITHM's instructions retired per second, every instruction priced at the rate the
        # device sustains for its class at this kernel's occupancy (measured, see `pricing`); peak = issue cycles the chip has per second
        'achieved': alg_cycles_per_ray * rays / k_s / 1e9, 'peak': N_SIMD * CLOCK_HZ / 1e9, 'unit': 'G SIMD issue cycles/s',
        'frac': alg_floor_ms / kernel_ms, 'traffic': traffic, 'kernel_ms': kernel_ms,
        'algorithmic_valu_floor': {
            'lane_ops_per_ray': {k: alg[k] for k in ('full', 'half', 'quarter')}, 'mfma_per_ray': alg['mfma_f16'],
            'issue_cycles_per_ray': alg_cycles_per_ray, 'floor_ms': alg_floor_ms, 'frac': alg_floor_ms / kernel_ms,
            'pricing': {'simd_cycles_per_wave64_instruction': price, 'source': price_src,
                        'note': 'measured at 4 waves per SIMD (the kernel\'s occupancy); MFMA and VALU issue cycles of co-resident waves add '
                                f'(probe: mfma_f16+8fma); {N_SIMD} SIMDs at {CLOCK_HZ / 1e9} GHz'},
            'frac_if_every_valu_cost_2_cycles': ((alg['full'] + alg['half']) / 64 * 2 + alg['quarter'] / 64 * 8 + alg['mfma_f16'] * 8)
                                                * rays / (N_SIMD * CLOCK_HZ) * 1e3 / kernel_ms,
            # continuity with the lines of rounds 1-3 and of the first half of round 4, whose table charged softplus the overflow-safe form
            # the kernel executed then (one more full-rate add and one half-rate max per hidden unit and sample)
            'frac_on_the_table_before_the_short_softplus': (alg_cycles_per_ray + (s + f) * 64 * (price['full'] + price['half']) / 64)
                                                           * rays / (N_SIMD * CLOCK_HZ) * 1e3 / kernel_ms},
        'l1_gather_floor': {'floor_ms': gather_floor_ms, 'frac': gather_floor_ms / kernel_ms,
                            'note': f'12 taps x 128 B per sample at the measured {GATHER_BYTES_PER_CYCLE_PER_CU:.0f} B/cycle/CU of the vector-memory path '
                                    '(probe `gather`): the second-busiest unit, overlapped with the vector issue'},
        'fp32_matrix_yardstick': {'TFLOPs': flops / k_s / 1e12, 'peak_TFLOPs': PEAK_FP32_MFMA_TFLOPS, 'frac': flops / k_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                  'note': 'algorithmic MLP FLOPs (8320 per sample) / kernel time against the fp32-input MFMA peak: a speed yardstick '
                                          '(the results are fp32-grade), NOT utilisation of a pipe the kernel uses'},
        'executed_f16_mfma': {'TFLOPs': 3 * samples * (2 * 32 * 64 + 2 * 64 * 32) / k_s / 1e12, 'peak_TFLOPs': PEAK_F16_MFMA_TFLOPS,
                              'frac': 3 * samples * (2 * 32 * 64 + 2 * 64 * 32) / k_s / 1e12 / PEAK_F16_MFMA_TFLOPS},
        'hbm': {'algorithmic_bytes_per_launch': hbm_bytes_per_call(n_items, rays, s, f, plane),
                'algorithmic_GBs': hbm_bytes_per_call(n_items, rays, s, f, plane) / k_s / 1e9,
                'frac_of_8TBs': hbm_bytes_per_call(n_items, rays, s, f, plane) / k_s / 1e9 / PEAK_HBM_GBS},
        'l2_gather': {'algorithmic_bytes_per_launch': samples * GATHER_BYTES_PER_SAMPLE,
                      'algorithmic_GBs': samples * GATHER_BYTES_PER_SAMPLE / k_s / 1e9,
                      'algorithmic_frac_of_34_5TBs': samples * GATHER_BYTES_PER_SAMPLE / k_s / 1e9 / PEAK_L2_GBS},
        'note': 'Bound by vector-instruction issue (HBM 3 % of peak, L2 hit 98 %).  frac = SIMD cycles the algorithm\'s own instructions need at '
                'the measured per-class issue rates / SIMD cycles of the kernel; measured_issue prices the EXECUTED instruction count.',
    }
    if pmc:
        valu, n_mfma = pmc.get('SQ_INSTS_VALU'), pmc.get('SQ_INSTS_MFMA')
        if valu and n_mfma:
            # SQ_INSTS_VALU includes the MFMAs.  The algorithm's own instructions are priced by class; what the kernel executes beyond
            # them (addressing, selects, conversions, LDS hand-offs, scalar-wave bookkeeping: integer / compare / select work) at the half rate
            alg_insts = (alg['full'] + alg['half'] + alg['quarter']) / 64 * rays
            extra = max(0.0, valu - n_mfma - alg_insts)
            cyc = alg_cycles_per_ray * rays + extra * price['half']
            floor_ms = cyc / (N_SIMD * CLOCK_HZ) * 1e3
            out['measured_issue'] = {'source': pmc_source, 'insts_valu_per_ray': (valu - n_mfma) / rays, 'insts_mfma_per_ray': n_mfma / rays,
                                     'issue_ms_of_executed_instructions': floor_ms, 'issue_busy_frac': floor_ms / kernel_ms,
                                     'algorithmic_over_executed_valu': alg_insts / (valu - n_mfma)}
        if pmc.get('TCP_TCC_READ_REQ_sum'):
            b = pmc['TCP_TCC_READ_REQ_sum'] * 128
            out['l2_gather'].update({'counter_bytes_per_launch': b, 'counter_GBs': b / k_s / 1e9, 'counter_frac_of_34_5TBs': b / k_s / 1e9 / PEAK_L2_GBS,
                                     'source': pmc_source + ' TCP_TCC_READ_REQ x 128 B'})
        if pmc.get('SQ_LDS_BANK_CONFLICT') and pmc.get('SQ_ACTIVE_INST_LDS'):
            out['lds_conflict_over_active'] = pmc['SQ_LDS_BANK_CONFLICT'] / pmc['SQ_ACTIVE_INST_LDS']
    return out


def latest_pmc():
    """(counters, 'profiles/<file>') of the latest committed PMC pass of this command, or (None, None)."""
    import re
    pdir = os.path.join(ROOT, 'profiles')
    try:
        named = [(re.fullmatch(r'r(\d+)_final(\d*)_pmc\.json', f), f) for f in os.listdir(pdir)]
        latest = sorted((int(m.group(1)), int(m.group(2) or 0), f) for m, f in named if m)[-1][2]
        doc = json.load(open(os.path.join(pdir, latest)))
        # profile_head: the commit the counters were collected on (tools/prof_collect.py records it), so that a reader can see when a
        # quoted counter is older than the kernel this run timed
        return doc['counters_mean_per_launch'], f"profiles/{latest} (collected at {doc.get('head', 'an unrecorded commit')})"
    except Exception:
        return None, None


def build_head():
    """The commit the running library was built from (g-nerf_amd/gnerf_hip/BUILD_HEAD, written by csrc/build.sh), or None."""
    try:
        return open(os.path.join(ROOT, 'g-nerf_amd', 'gnerf_hip', 'BUILD_HEAD')).read().strip() or None
    except OSError:
        return None


def hbm_bytes_per_call(n_items, rays, s, f, plane):
    planes = n_items * 3 * 32 * plane * plane * 4
    return planes + rays * 3 * 4 * 2 + rays * (s + f) * 4 + rays * 34 * 4 + (64 * 32 + 64 + 33 * 64 + 33) * 4


def _scene(dev, seed, n_items=N_ITEMS, plane=PLANE):
    """Synthetic inputs of BASELINE.md section 2: randn planes, default-init OSGDecoder (W~N(0,1), b=0, gains
    folded), cameras LookAtPoseSampler.sample(3.14/2, 3.14/2, radius=2.7), FFHQ intrinsics."""
    import math
    gen = torch.Generator().manual_seed(seed)
    planes = torch.randn(n_items, 3, 32, plane, plane, generator=gen)
    w1 = torch.randn(64, 32, generator=gen) / math.sqrt(32)
    w2 = torch.randn(33, 64, generator=gen) / math.sqrt(64)
    dec = (w1, torch.zeros(64), w2, torch.zeros(33))
    # camera_utils.py:89-106,155-174 for yaw = pitch = 3.14/2, radius 2.7 (values pinned by tests/golden/camera.npz)
    theta, phi, r = 3.14 / 2, 3.14 / 2, 2.7
    org = torch.tensor([r * math.sin(phi) * math.cos(math.pi - theta), r * math.cos(phi), r * math.sin(phi) * math.sin(math.pi - theta)])
    fwd = -org / org.norm()
    up = torch.tensor([0.0, 1.0, 0.0])
    right = -torch.linalg.cross(up, fwd)
    right = right / right.norm()
    up2 = torch.linalg.cross(fwd, right)
    up2 = up2 / up2.norm()
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up2, fwd, org
    c2w = c2w[None].repeat(n_items, 1, 1)
    intr = torch.tensor([[4.2647, 0, 0.5], [0, 4.2647, 0.5], [0, 0, 1]])[None].repeat(n_items, 1, 1)
    return planes.to(dev), tuple(t.to(dev) for t in dec), c2w.to(dev), intr.to(dev)


def _cpu_oracle_rays_per_s(threads, n_items, seconds_budget, max_passes=3):
    """rays/s of the CPU oracle on config 2 restricted to `n_items` items, `threads` torch threads: median of the timed passes after one
    warm-up (or the warm-up itself when the budget ends there)."""
    from oracle import render_ref as R
    torch.manual_seed(0)
    torch.set_num_threads(threads)
    planes, dec, c2w, intr = _scene(torch.device('cpu'), 0, n_items=n_items)
    o, d = R.make_rays(c2w, intr, RES)
    rays = n_items * RES * RES
    opts = dict(depth_resolution=S_COARSE, depth_resolution_importance=S_FINE, ray_start=RAY_START, ray_end=RAY_END,
                box_warp=BOX_WARP, clamp_mode='softplus')
    times = []
    t_all = time.time()
    for it in range(1 + max_passes):
        nc, nf = torch.rand(n_items, RES * RES, S_COARSE), torch.rand(rays, S_FINE)
        t0 = time.time()
        with torch.no_grad():
            R.render(planes, dec, o, d, opts, nc, nf)
        times.append(time.time() - t0)
        if time.time() - t_all > seconds_budget:
            break
    timed = sorted(times[1:] or times)
    return {'value': rays / timed[len(timed) // 2], 'threads': threads, 'items': n_items, 'rays': rays, 'passes_timed': len(times[1:]),
            'warm_up_was_the_measurement': len(times) == 1}


def host_cpu_topology():
    """Logical CPUs, physical cores (distinct (physical id, core id) pairs of /proc/cpuinfo), the CPUs this process may run on
    (affinity) and the cgroup's CPU quota in cores (cpu.max), each None where the host does not say."""
    logical = os.cpu_count() or 1
    phys = None
    try:
        pairs, cur = set(), {}
        for ln in open('/proc/cpuinfo'):
            if ':' in ln:
                k, v = [x.strip() for x in ln.split(':', 1)]
                cur[k] = v
            elif cur:
                pairs.add((cur.get('physical id', '0'), cur.get('core id', cur.get('processor'))))
                cur = {}
        if cur:
            pairs.add((cur.get('physical id', '0'), cur.get('core id', cur.get('processor'))))
        phys = len(pairs) or None
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    quota = None
    for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(f).read().split()
            if f.endswith('cpu.max'):
                if txt[0] != 'max':
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return {'logical_cpus': logical, 'physical_cores': phys, 'affinity_cpus': affinity, 'cgroup_quota_cores': quota}


def _cpu_worker(threads, limit_s):
    """The oracle on ONE item of config 2 with `threads` torch threads, in a child process that cannot see a GPU, under a hard limit."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', str(threads)], capture_output=True, text=True,
                           timeout=limit_s, env=dict(os.environ, HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES=''))
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    except subprocess.TimeoutExpired:
        return {'value': None, 'threads': threads, 'note': f'one item of the batch did not finish inside {limit_s:.0f} s'}
    except Exception as e:
        return {'value': None, 'threads': threads, 'note': f'{type(e).__name__}: {e}'[:200]}


def cpu_baseline(seconds_budget=30.0):
    """The CPU oracle (a port of the reference's PyTorch path, pinned to it by tests/golden) timed on this host.
    1. A thread-count SWEEP on one item of config 2's batch, each count in a child process with a hard limit: the cores this process is
       actually given (cgroup quota / affinity), the physical cores, twice that, 32, in one pass each after a warm-up.
    2. `value`: the WHOLE batch of config 2 (4 items x 128x128 rays, 48+48 samples, 256x256x32 planes) in this process at the sweep's
       best count -- the oracle is bandwidth-bound gather code, torch's CPU ops get slower once threads outnumber the cores given.
    3. `all_cores`: BASELINE.md section 2's literal setting, torch.set_num_threads(os.cpu_count()), on one item, child process, 45 s
       limit (on a box whose CPU share is far below os.cpu_count() that oversubscribes and times out; reported as such)."""
    topo = host_cpu_topology()
    host = topo['logical_cpus']
    given = topo['cgroup_quota_cores'] or topo['affinity_cpus'] or host
    cand = {int(max(1, min(host, round(given)))), 32 if host >= 32 else host}
    if topo['physical_cores']:
        cand |= {min(host, topo['physical_cores']), min(host, 2 * topo['physical_cores'])}
    sweep = [_cpu_worker(t, 40.0) for t in sorted(cand)]
    ok = [r for r in sweep if r.get('value')]
    best_threads = max(ok, key=lambda r: r['value'])['threads'] if ok else max(1, min(host, 32))
    main = _cpu_oracle_rays_per_s(best_threads, N_ITEMS, seconds_budget * 0.6)
    all_cores = next((r for r in sweep if r['threads'] == host), None) or _cpu_worker(host, 45.0)
    return {'value': main['value'], 'unit': 'rays/s', 'cores': main['threads'], 'host_cpu_count': host, 'kind': 'port', 'host': topo,
            'thread_sweep_one_item': [{'threads': r['threads'], 'value': r.get('value'), 'note': r.get('note')} for r in sweep],
            'all_cores': all_cores,
            'sample': f'config 2 whole batch ({main["rays"]} rays, 48+48 samples, 4x3x32x256x256 planes); median of {main["passes_timed"]} pass(es) after 1 '
                      f'warm-up, torch {torch.__version__} CPU fp32, {main["threads"]} threads = the best of a sweep over {sorted(cand)} threads on one item '
                      f'(child processes, 40 s limit each); host: {host} logical CPUs, {topo["physical_cores"]} physical cores, '
                      f'{given:g} given to this process; all_cores = os.cpu_count() threads (BASELINE.md section 2) on one item, child process'}


def gen_videos_secondary(rank, world, dev, n_frames=240, flows=('fast', 'reference')):
    """BASELINE's second metric, frames/sec of gen_videos (config 4): the 240-frame orbit of gen_videos.py:154-171 sharded in
    contiguous blocks over the ranks (30 frames per GPU at 8 GPUs), 64x64 rays x (96+96) samples per frame (the CLI's doubled
    sampling), cached backbone, superresolution to 512x512 in fp16, uint8 frames, ONE gather of the frames to rank 0 at the end (RCCL).
    Generator: random-init FFHQ configuration (gnerf_generator.Generator -- the reference's layer graph around this repo's
    renderer and ops; there is no reference tree or checkpoint on the GPU box).  One untimed warm-up frame (MIOpen's
    per-shape kernel search), then the orbit eagerly and replayed from a captured HIP graph, in TWO flows:
      fast       gnerf_generator's own path: csrc/modconv.hip around the convolutions, channels_last planes from the producer kernel
      reference  the layer code a G-NeRF checkout runs (GNERF_MODCONV_FAST=0: modulated_conv2d's PyTorch ops, convolutions through
                 the overlay's torch_utils.ops.conv2d_resample / fma, native bias_act / upfirdn2d, fused renderer) with NCHW planes
    Returns a dict (all ranks); `value` is the fast flow's better number, `reference_flow_value` the other flow's."""
    from torch_utils import custom_ops
    custom_ops.verbosity = 'none'                     # stdout carries exactly one JSON line
    import gnerf_harness as H
    import gen_videos_mi355x as gv
    import gnerf_generator as GG
    if world > 1:
        import torch.distributed as dist
    out = {}
    searched = H.configure_backend()                  # MIOpen solver search per shape, in the warm-up frames (GNERF_MIOPEN_FIND=0: off)
    with torch.no_grad():
        G = gv.build_random_generator(0, dev)
        z = torch.randn(1, G.z_dim, generator=torch.Generator().manual_seed(1)).to(dev)
        last = G.backbone.synthesis.b256
        if world > 1:
            # every convolution shape of the run (both flows, one camera per call and ORBIT_VIEWS), searched by rank 0 alone; the other ranks
            # have not issued a convolution yet (building the generator only allocates and initialises) -- see one_solver_search
            def warm_all():
                for f in flows:
                    GG._MODCONV_FAST = f == 'fast'
                    last.emit_channels_last = f == 'fast'
                    gv.render_orbit(G, z, n_frames, 64, dev, rank=0, world=n_frames, double_depth=(f == 'fast'))
                    if f == 'fast':
                        gv.render_orbit(G, z, ORBIT_VIEWS, 64, dev, double_depth=False, frames_per_call=ORBIT_VIEWS)
                torch.cuda.synchronize()
            t_search = time.perf_counter()
            one_solver_search(rank, world, warm_all, dist.barrier, sync=torch.cuda.synchronize)
            out['solver_search_s'] = time.perf_counter() - t_search
        for flow, fast in [(f, f == 'fast') for f in flows]:
            GG._MODCONV_FAST = fast
            last.emit_channels_last = fast
            gv.render_orbit(G, z, n_frames, 64, dev, rank=0, world=n_frames, double_depth=(flow == 'fast'))      # warm-up: frame 0 (the first also sets 96+96)
            # the frame's launch sequence is captured once per generator, latent and flow, outside the timed orbit, like the warm-up frame
            program = gv.FrameProgram(G, gv.orbit_latents(G, z, dev), 64, dev)
            runs = [('eager', False, 1), ('hip_graph', True, 1)]
            if fast:
                # ORBIT_VIEWS cameras per synthesis call: the renderer takes them as views of the one latent's planes in one launch
                # (each with the draws and the depth clamp of a call of its own: tests/test_gpu_parity.py::
                # test_views_of_one_item_equal_separate_calls), the superresolution as a batch
                gv.render_orbit(G, z, ORBIT_VIEWS, 64, dev, double_depth=False, frames_per_call=ORBIT_VIEWS)         # warm-up of the batch-k shapes
                runs.append(('eager_views', False, ORBIT_VIEWS))
                # ... and the same ORBIT_VIEWS-camera call captured once into a HIP graph (its own FrameProgram: the batch size is baked in)
                program_k = gv.FrameProgram(G, gv.orbit_latents(G, z, dev), 64, dev, batch=ORBIT_VIEWS)
                runs.append(('hip_graph_views', True, ORBIT_VIEWS))
            for name, use_graph, k in runs:
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                frames, _, _ = gv.render_orbit(G, z, n_frames, 64, dev, rank, world, double_depth=False,
                                               program=(program_k if k > 1 else program) if use_graph else None, frames_per_call=k)
                full = H.gather_frames(frames, n_frames)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                out[flow, name] = n_frames / H.max_over_ranks(time.perf_counter() - t0, dev)
                if rank == 0:
                    assert full.shape == (n_frames, 512, 512, 3) and full.dtype == torch.uint8
            del program
            if fast:
                del program_k
        GG._MODCONV_FAST = os.environ.get('GNERF_MODCONV_FAST', '1') != '0'
        last.emit_channels_last = True
        sec_roofline = None
        if rank == 0:
            try:
                sec_roofline = secondary_roofline(dev, G, z, ORBIT_VIEWS)
            except Exception as e:
                sec_roofline = {'error': f'{type(e).__name__}: {e}'[:300]}
        # the backbone pass every rank runs once per orbit before its frames (ws is constant, gen_videos.py:150): the serial term of
        # config 4's scaling -- 240 frames on one GPU against (backbone + 240 / N frames + gather) on N
        ws = gv.orbit_latents(G, z, dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = []
        for _ in range(4):
            e0.record()
            G.backbone.synthesis(ws, noise_mode='const')
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        backbone_ms = sorted(ms[1:])[1]
    backbone_all = [backbone_ms]
    if world > 1:
        t = torch.tensor([backbone_ms], device=dev, dtype=torch.float64)
        allb = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allb, t)
        backbone_all = [float(x[0]) for x in allb]
    best = max(('eager', 'hip_graph', 'eager_views', 'hip_graph_views'), key=lambda k: out['fast', k])
    return {'metric': 'frames/sec gen_videos', 'value': out['fast', best], 'unit': 'frames/s', 'value_is': 'fast flow, ' + best,
            'eager_value': out['fast', 'eager'], 'hip_graph_value': out['fast', 'hip_graph'],
            'eager_views_value': out['fast', 'eager_views'], 'hip_graph_views_value': out['fast', 'hip_graph_views'], 'views_per_call': ORBIT_VIEWS, 'miopen_solver_search': searched,
            'backbone_ms_per_rank': backbone_all, 'roofline': sec_roofline,
            'solver_search': ({'shared': True, 'rank0_then_copy_s': out.get('solver_search_s')} if world > 1 else {'shared': False}),
            'reference_flow_value': max(out['reference', 'eager'], out['reference', 'hip_graph']) if 'reference' in flows else None,
            'reference_flow_eager_value': out.get(('reference', 'eager')), 'reference_flow_hip_graph_value': out.get(('reference', 'hip_graph')),
            'flows_measured': list(flows), 'n_gpus': world,
            'workload': f'config 4: {n_frames}-frame orbit sharded over {world} GPU(s), 64x64 rays x (96+96) samples, cached backbone, SR to '
                        '512x512 fp16, uint8 frames, one gather to rank 0; random-init FFHQ-config generator; hip_graph = HIP-graph replay of the '
                        'per-frame sequence (captured once, before the timed orbit), eager = plain launches (backbone pass included), '
                        'eager_views = plain launches with views_per_call cameras per synthesis call, hip_graph_views = that call captured once and replayed; '
                        'fast = this repo\'s generator path (modconv kernels, channels_last planes), reference_flow = the reference\'s layer code '
                        '(PyTorch-op modulation, conv2d_resample / fma / bias_act / upfirdn2d from the overlay, NCHW planes): what a G-NeRF checkout gets'}


F16_MATRIX_PEAK_PFLOPS = 2.5          # dense f16 MFMA peak of one MI355X at 2.4 GHz (MI355X_MICROARCH.md; AMD's headline 5 PF includes 2:1 sparsity)


def secondary_roofline(dev, G, z, views):
    """Round 6: the roofline of the SECONDARY metric's dominant kernel in the driver's own line.  Half of an orbit frame's GPU time is the
    superresolution's 3x3 convolutions on csrc/conv3x3.hip (profiles/r05_orbit_fast_views8_summary.json), a matrix-core kernel: it is timed
    here, live, with HIP events on the two hot shapes of the plain form and the larger one of the transposed form, at the batch the orbit's
    best run uses (`views` cameras per call) -- `frac` = PFLOP/s over the 2.5 PFLOP/s f16 peak.  Then ONE more pass over 30 frames under
    torch's profiler (roctracer) splits a frame's GPU time by kernel family; it is skipped (null) when an outer profiler owns the tracer."""
    import gnerf_hip
    import gen_videos_mi355x as gv
    out = {'bound': 'mfma', 'peak': F16_MATRIX_PEAK_PFLOPS, 'unit': 'PFLOP/s', 'dtype': 'f16 operands, fp32 accumulate', 'batch': views, 'kernels': []}
    g = torch.Generator().manual_seed(3)

    def timed(fn, reps=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        return best
    for kind, (c, o, h, w) in (('conv3x3_epilogue', (128, 128, 512, 512)), ('conv3x3_epilogue', (256, 256, 256, 256)), ('conv_transpose3x3_s2', (256, 128, 256, 256))):
        x = (torch.randn(views, c, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(o, c, 3, 3, generator=g) / (3 * c ** 0.5)).to(dev)
        if kind == 'conv3x3_epilogue':
            wpk = gnerf_hip.pack_conv3x3_weights(wt)
            sc, nx, bias = (torch.rand(views, o, generator=g) + 0.5).to(dev), (torch.rand(views, o, generator=g) + 0.5).to(dev), torch.zeros(o, device=dev)
            ms = timed(lambda: gnerf_hip.conv3x3_epilogue(x, wpk, bias, scale=sc, next_scale=nx, gain=2 ** 0.5, clamp=256.0))
            flop = 2.0 * views * h * w * o * c * 9
        else:
            wpk = gnerf_hip.pack_conv_transpose3x3_weights(wt)
            ms = timed(lambda: gnerf_hip.conv_transpose3x3_s2(x, wpk))
            flop = 2.0 * views * h * w * o * c * 9                   # every input pixel meets each of the nine taps once
        pf = flop / (ms * 1e-3) / 1e15
        out['kernels'].append({'kernel': kind, 'shape': {'n': views, 'cin': c, 'cout': o, 'h': h, 'w': w}, 'ms': ms, 'achieved': pf, 'frac': pf / F16_MATRIX_PEAK_PFLOPS,
                               'algorithmic_GFLOP': flop / 1e9})
        del x
    out['achieved'] = out['kernels'][0]['achieved']
    out['frac'] = out['kernels'][0]['frac']
    out['kernel'] = 'conv3x3_epilogue_kernel<0, ...> (csrc/conv3x3.hip), 128 -> 128 @ 512^2'
    # a frame's GPU time by kernel family
    fam = None
    if not (any(k.startswith(('ROCP_', 'ROCPROF')) or k == 'HSA_TOOLS_LIB' for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')):
        try:
            from torch.profiler import profile, ProfilerActivity
            n_frames = 32 // views * views
            gv.render_orbit(G, z, n_frames, 64, dev, double_depth=False, frames_per_call=views)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                gv.render_orbit(G, z, n_frames, 64, dev, double_depth=False, frames_per_call=views)
                torch.cuda.synchronize()
            fam = {}
            rules = (('conv3x3 (csrc/conv3x3.hip)', ('conv3x3_epilogue_kernel',)), ('fused renderer', ('render_kernel', 'clamp_depth', 'make_rays')),
                     ('blur / upfirdn2d', ('upfirdn', 'blur')), ('modulated-convolution surroundings, bias_act, ToRGB, uint8', ('modconv', 'bias_act', 'torgb', 'scale_channels', 'to_uint8', 'upsample2x', 'modulate_weights', 'normalise')),
                     ('MIOpen / rocBLAS', ('miopen', 'Cijk', 'igemm', 'naive_conv', 'ck::', 'gemm', 'Conv', 'batched_transpose')), ('uniform draws', ('philox', 'distribution', 'rand')))
            total = 0.0
            for ev in prof.key_averages():
                t = float(getattr(ev, 'device_time_total', 0.0) or getattr(ev, 'cuda_time_total', 0.0) or 0.0)
                if t <= 0:
                    continue
                name = ev.key
                key = next((f for f, pats in rules if any(p_ in name for p_ in pats)), 'torch elementwise / other')
                fam[key] = fam.get(key, 0.0) + t / n_frames
                total += t / n_frames
            fam = {k: round(v, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])}
            fam['total'] = round(total, 2)
        except Exception as e:
            fam = {'error': f'{type(e).__name__}: {e}'[:200]}
    out['gpu_time_us_per_frame'] = fam
    out['gpu_time_source'] = f'torch.profiler (roctracer), {views} cameras per synthesis call, eager launches, 32 frames' if fam and 'error' not in fam else None
    return out


def realistic_planes_step(dev, c2w, intr, steps, rank=0):
    """Config 2 a second time on planes a generator actually produces ("optionally a second run with real backbone
    output to exercise realistic density"): planes = the random-init FFHQ StyleGAN2 backbone's output for 4 latents (triplane.py:69-77;
    channels_last from the plane producer, max |planes| with them), decoder = the generator's default-init OSGDecoder, same cameras,
    same step as the headline's producer-layout form (make_rays + 2 torch.rand + fused render + depth clamp).  Says which decoder
    arithmetic the device-side check picks for such planes."""
    import gnerf_hip
    import gen_videos_mi355x as gv
    from training.volumetric_rendering import renderer as RM
    with torch.no_grad():
        G = gv.build_random_generator(0, dev)
        z = torch.randn(N_ITEMS, G.z_dim, generator=torch.Generator().manual_seed(11 + rank)).to(dev)
        ws = G.mapping(z, torch.zeros(N_ITEMS, 25, device=dev))
        img = G.backbone.synthesis(ws, noise_mode='const')                       # [4,96,256,256], channels_last memory on the fast path
        planes5 = img.view(N_ITEMS, 3, 32, PLANE, PLANE)
        nhwc, amax = G.renderer._planes_nhwc(planes5)
        dec = G.renderer._decoder_cache(RM._osg_decoder_weights(G.decoder))
        stats = {'absmax': float(planes5.abs().max()), 'std': float(planes5.std()), 'mean': float(planes5.mean())}
        kw = dict(depth_resolution=S_COARSE, depth_resolution_importance=S_FINE, ray_start=RAY_START, ray_end=RAY_END, box_warp=BOX_WARP, image_width=RES)

        def one(ev=None):
            o, d = gnerf_hip.make_rays(c2w, intr, RES)
            nc = torch.rand([N_ITEMS, RES * RES, S_COARSE, 1], device=dev)
            nf = torch.rand(N_ITEMS * RES * RES, S_FINE, device=dev)
            if ev:
                ev[0].record()
            out = gnerf_hip.render_forward(nhwc, N_ITEMS, dec, o, d, nc, nf, planes_absmax=amax, **kw)
            if ev:
                ev[1].record()
            return out
        for _ in range(10):
            out = one()
        torch.cuda.synchronize()
        choice = gnerf_hip.last_mlp_choice(dev)
        regions, calls = [], []
        for _ in range(5):
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                out = one(evs[i] if i % 5 == 0 else None)
            torch.cuda.synchronize()
            regions.append(time.perf_counter() - t0)
            calls.append(sum(a.elapsed_time(b) for a, b in evs[::5]) / len(evs[::5]))
        el = sorted(regions)[2]
        wsum = out[2]
        return {'workload': 'config 2 on backbone-output planes: random-init FFHQ StyleGAN2 backbone (4 latents, noise_mode const) -> [4,96,256,256] channels_last '
                            'planes + max |planes| from the plane producer, the generator\'s default-init OSGDecoder, same cameras and sampling as the headline',
                'value': N_ITEMS * RES * RES * steps / el, 'unit': 'rays/s', 'ms_per_step': 1e3 * el / steps, 'render_call_ms': sorted(calls)[2],
                'last_mlp_choice': choice, 'planes': stats,
                'opacity': {'mean_weight_sum': float(wsum.mean()), 'frac_rays_weight_sum_above_0.5': float((wsum > 0.5).float().mean())}}


ORBIT_VIEWS = 8                   # cameras per synthesis call of the batched orbit (tools/orbit_marked.py --frames-per-call 4 / 6 / 8 / 12 -> 2304 / 2354 / 2387 / 2298
                                  # frames/s with plain launches on the final library, profiles/r05_orbit_views_k.jsonl; round 3, on MIOpen's convolutions: 2 / 4 / 8 -> 1235 / 1357 / 1327)
PEAK_ATOMIC_GBS = 1330.0          # chip-wide float-atomic rate: 20.8 G 64-byte requests/s on this device (profiles/r04_atomic_scope_probe.txt; MI355X_MICROARCH.md gives 1.26-1.36 TB/s of added bytes)
FLOP_BWD_PER_SAMPLE = 3 * FLOP_MLP_PER_SAMPLE       # one forward recomputation + dX / dW products of both layers (fp32 MFMA)


def backward_times(dev, planes_cl, dec, c2w, intr, reps=5):
    """Renderer backward at config 2, timed with HIP events on the launch stream around gnerf_render_backward:
    the whole call (staged scatter: pass 1 render_bwd_kernel + pass 2 plane_scatter_kernel), the single-pass form, and the
    decoder-only request (= pass 1's compute without any plane scatter).  Returns the `roofline_backward` object."""
    import gnerf_hip
    o, d = gnerf_hip.make_rays(c2w, intr, RES)
    rays = N_ITEMS * RES * RES
    nc = torch.rand([N_ITEMS, RES * RES, S_COARSE, 1], device=dev)
    nf = torch.rand(rays, S_FINE, device=dev)
    g = [torch.randn(N_ITEMS, RES * RES, k, device=dev) for k in (32, 1, 1)]
    amax = gnerf_hip.planes_absmax(planes_cl)
    kw = dict(depth_resolution=S_COARSE, depth_resolution_importance=S_FINE, ray_start=RAY_START, ray_end=RAY_END, box_warp=BOX_WARP, image_width=RES)

    def timed(**extra):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ms = []
        for i in range(reps + 1):
            e0.record()
            gnerf_hip.render_backward(planes_cl, N_ITEMS, dec, o, d, nc, nf, *g, planes_absmax=amax, **kw, **extra)
            e1.record()
            torch.cuda.synchronize()
            if i:
                ms.append(e0.elapsed_time(e1))           # includes zero-filling the gradient buffers (100 MB memset, ~15 us)
        return sorted(ms)[len(ms) // 2]

    staged, direct, dec_only = timed(), timed(staged_scatter=False), timed(need_planes=False)
    os.environ['GNERF_BWD_KERNEL'] = 'wave'            # the one-wave-per-ray kernel (every shape outside the pipelined kernels' range; rounds 1-3's only form)
    try:
        staged_wave = timed()
    finally:
        os.environ.pop('GNERF_BWD_KERNEL', None)
    samples = rays * (S_COARSE + S_FINE)
    flops = samples * FLOP_BWD_PER_SAMPLE
    atom_bytes = samples * 12 * 32 * 4                  # one 4-byte add per tap and channel: what grid_sampler_2d_backward issues too
    # the kernels of the staged form separately: from the committed rocprofv3 profile of the same call (tools/prof_bwd.sh).  Round 4:
    # pass 1 = render_kernel_pipe_bwd (ray level, on the forward pipeline) + render_bwd_tiles_kernel (per 16-sample tile); pass 2 unchanged
    passes = None
    try:
        pdir = os.path.join(ROOT, 'profiles')
        name = sorted(f for f in os.listdir(pdir) if f.endswith('_backward_profile.json'))[-1]
        doc = json.load(open(os.path.join(pdir, name)))
        prof = doc['staged']
        k1 = next((v for k, v in prof.items() if k.startswith('render_kernel_pipe_bwd')), None)
        k2 = prof.get('render_bwd_tiles_kernel')
        p1_ms = (k1['avg_ms'] + k2['avg_ms']) if (k1 and k2) else prof['render_bwd_kernel<true>']['avg_ms']
        bins = {k: v['avg_ms'] * (v.get('calls_per_launch', 1)) for k, v in prof.items() if k.startswith('bin_')}
        passes = {'source': f'profiles/{name} (rocprofv3 --kernel-trace --stats and --pmc TCC_EA0_ATOMIC_sum of tools/bench_bwd.py 4 128; collected at {doc.get("head", "an unrecorded commit")})',
                  'pass1_ms': p1_ms, 'pass1_ray_level_kernel_ms': k1 and k1['avg_ms'], 'pass1_tile_kernel_ms': k2 and k2['avg_ms'],
                  'pass1_fp32_mfma_frac': flops / (p1_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
        if bins:                # round 5: the binned scatter (csrc/scatter_binned.inl): no global float atomics
            p2_ms = sum(bins.values())
            rows_bytes = samples * 3 * 128 + samples * 3 * 24 + 3 * 32 * 4 * N_ITEMS * PLANE * PLANE * 2       # dX rows read per (sample, plane) + 24-byte records + the gradient read and written
            passes.update({'pass2_binned_scatter_ms': p2_ms, 'pass2_kernels_ms': bins,
                           'pass2_hbm': {'algorithmic_bytes': rows_bytes, 'GBs': rows_bytes / (p2_ms * 1e-3) / 1e9, 'frac_of_8TBs': rows_bytes / (p2_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                         'note': 'every dX row is read once per plane it contributes to (3 x 128 B per sample) next to its 24-byte record; the '
                                                 'sums are formed in LDS in 64-bit fixed point (ds_add_u64) and leave as plain stores'}})
            srt = doc.get('sorted', {}).get('plane_scatter_kernel')
            if srt:
                passes['pass2_sorted_form_ms'] = srt['avg_ms']
                passes['pass2_sorted_form_atomic_requests'] = srt.get('TCC_EA0_ATOMIC_sum')
        else:
            p2 = prof['plane_scatter_kernel']
            passes.update({'pass2_plane_scatter_kernel_ms': p2['avg_ms'], 'pass2_atomic_requests': p2['TCC_EA0_ATOMIC_sum'],
                           'pass2_atomic_frac': p2['TCC_EA0_ATOMIC_sum'] * 64 / (p2['avg_ms'] * 1e-3) / 1e9 / PEAK_ATOMIC_GBS})
    except Exception:
        pass
    out = {'workload': 'gnerf_render_backward at config 2 (4 x 128^2 rays, 48+48 samples), planes in the producer layout',
           'call_ms': {'staged_scatter': staged, 'single_pass': direct, 'decoder_gradients_only': dec_only, 'staged_scatter_one_wave_per_ray_kernel': staged_wave},
           'ratio_to_forward_kernel': None,
           'pass1_fp32_mfma': {'algorithmic_TFLOPs': flops / (dec_only * 1e-3) / 1e12, 'peak_TFLOPs': PEAK_FP32_MFMA_TFLOPS,
                               'frac': flops / (dec_only * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                               'note': f'{FLOP_BWD_PER_SAMPLE} FLOP per sample (forward recomputation + the four gradient products) over the decoder-only call'},
           'scatter_atomics': {'algorithmic_GBs_single_pass': atom_bytes / (direct * 1e-3) / 1e9, 'peak_GBs': PEAK_ATOMIC_GBS,
                               'frac_single_pass': atom_bytes / (direct * 1e-3) / 1e9 / PEAK_ATOMIC_GBS,
                               'note': 'one fp32 atomic per tap and channel (147 KB per ray) against the chip-wide float-atomic rate; the staged form issues '
                                       '~3.8x fewer 64-byte requests (profiles/r0N_backward_profile.json: TCC_EA0_ATOMIC 149.2 M -> 39.9 M per launch)'}}
    if passes:
        out['pass_ms'] = passes
    return out


def visible_gpus_no_hip(sysfs='/sys/class/kfd/kfd/topology/nodes', dev_dir='/dev/dri'):
    """How many GPUs a child of this process could open, found WITHOUT a HIP / HSA call: the kernel driver's topology nodes with
    `simd_count > 0` (CPU nodes have 0) whose render node this user can open, capped by the *_VISIBLE_DEVICES lists.  None when the
    topology cannot be read (then nothing is refused here; the ranks find out themselves).  The self-launching parent must never touch
    the GPU runtime -- a process that has opened it may only start fresh children or exit -- and torch.cuda.device_count() stays
    outside HIP only while amdsmi answers; it falls through to hipGetDeviceCount() otherwise."""
    import re
    try:
        nodes = sorted(os.listdir(sysfs), key=lambda x: (len(x), x))
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = open(os.path.join(sysfs, node, 'properties')).read()
        except OSError:
            continue
        simd = re.search(r'^simd_count\s+(\d+)', props, re.M)
        if not simd or int(simd.group(1)) == 0:
            continue
        minor = re.search(r'^drm_render_minor\s+(\d+)', props, re.M)
        if minor and int(minor.group(1)) > 0 and os.path.isdir(dev_dir):
            if not os.access(os.path.join(dev_dir, f'renderD{minor.group(1)}'), os.R_OK | os.W_OK):
                continue                # a GPU of the host that this container / user was not given
        n += 1
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip()]))
    return n


def per_rank_miopen_env(rank, env=None):
    """MIOpen's user database and kernel cache in a directory of this rank's own (unless the caller set them): eight ranks that search
    solvers for the same shapes at the same time would otherwise read and rewrite ONE sqlite file / cache directory under $HOME."""
    import tempfile
    env = os.environ if env is None else env
    base = os.path.join(tempfile.gettempdir(), f'gnerf_miopen_{os.getuid()}', f'rank{rank}')
    for var, sub in (('MIOPEN_USER_DB_PATH', 'db'), ('MIOPEN_CUSTOM_CACHE_DIR', 'cache')):
        if var not in env:
            os.makedirs(os.path.join(base, sub), exist_ok=True)
            env[var] = os.path.join(base, sub)
    return env


def one_solver_search(rank, world, warm, barrier, env=None, sync=None):
    """MIOpen's solver search ONCE per node instead of once per rank (round 6).  Every rank has MIOpen directories of its own
    (per_rank_miopen_env), so left alone eight ranks run eight identical searches at the same time -- each of them a few dozen kernel
    compilations and timing runs on a host that gives a process 16 cores: the likeliest way for a first 8-GPU run to meet its time limit.
    Here rank 0 runs `warm` (the untimed warm-up frames: every convolution shape of the run) ALONE while the other ranks wait in
    `barrier` -- which they enter before they have issued a single convolution, i.e. before MIOpen has opened any database --, then its
    user database and kernel cache are copied into every other rank's directories (one host, plain files), the barrier releases, and the
    other ranks' `warm` finds every solver and every compiled kernel cached.  Directories the caller set by hand are left alone (a shared
    directory needs no copy).  The reference starts its per-GPU processes the same way and lets each search (train.py:40-56,104-111,
    training_loop.py:133,144).  Returns warm()'s value."""
    import shutil
    env = os.environ if env is None else env
    if world <= 1:
        return warm()
    result = None
    if rank == 0:
        result = warm()
        if sync is not None:
            sync()
        for var in ('MIOPEN_USER_DB_PATH', 'MIOPEN_CUSTOM_CACHE_DIR'):
            src = env.get(var)
            marker = os.sep + 'rank0' + os.sep
            if not src or marker not in src + os.sep or not os.path.isdir(src):
                continue                                    # not one of per_rank_miopen_env's directories
            for k in range(1, world):
                dst = (src + os.sep).replace(marker, f'{os.sep}rank{k}{os.sep}').rstrip(os.sep)
                shutil.copytree(src, dst, dirs_exist_ok=True)
    barrier()
    if rank != 0:
        result = warm()
    return result


def self_launch(n_ranks, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process -- which has made NO GPU call and makes none --
    starts N fresh children of this same script, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what
    torch.distributed.run would export; the reference's entry point spawns its per-GPU processes itself too, train.py:40-56,104-111),
    relays rank 0's single JSON line and exits non-zero when any rank did.  Under an outer torch.distributed.run (WORLD_SIZE set) this
    function is never reached.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               GNERF_BENCH_SELF_LAUNCHED='1')
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')               # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n_ranks) // n_ranks)))
    procs = []
    for r in range(n_ranks):
        e = per_rank_miopen_env(r, dict(env, RANK=str(r), LOCAL_RANK=str(r)))
        # rank 0's stdout is the result line (captured and relayed); the other ranks print nothing there by construction
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    # poll: once a rank has exited non-zero its peers may sit in a collective for ever -- give them 30 s, then end exactly those PIDs
    deadline = None
    while any(p.poll() is None for p in procs):
        if deadline is None and any(p.poll() not in (None, 0) for p in procs):
            deadline = time.time() + float(os.environ.get('GNERF_BENCH_PEER_GRACE_S', '30'))
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = b''.join(chunks)
    lines = [ln for ln in out0.decode(errors='replace').splitlines() if ln.strip()]
    if lines:
        sys.stdout.write(lines[-1] + '\n')
        sys.stdout.flush()
    if any(codes):
        print(f'bench.py: rank exit codes {codes}', file=sys.stderr)
        return max(1, next(c for c in codes if c))
    return 0 if lines else 1


def rank_identity(rank, world, dev):
    """What every rank reports about itself, gathered to all ranks (one small all_gather_object): the result line's `ranks` list
    proves which backend saw how many ranks on which devices."""
    me = {'rank': rank, 'pid': os.getpid(), 'device': str(dev)}
    if dev.type == 'cuda':
        pr = torch.cuda.get_device_properties(dev)
        me.update(device_name=pr.name, pci_bus_id=getattr(pr, 'pci_bus_id', None), uuid=str(getattr(pr, 'uuid', '')) or None)
    if world == 1:
        return [me]
    import torch.distributed as dist
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    return everyone


def stub_main(args, rank, world, result_fd):
    """--stub-step: the launcher, the rendezvous, the barriers, the max-over-ranks reduction and the per-rank report of this file with a
    step that does NO rendering (a 64-element CPU add), on the gloo backend, CPU only.  For the multi-rank rehearsal tests
    (tests/test_dist_cpu.py: 8 ranks here; no GPU needed) -- the line says `stub: true`, carries no rays/s value and is not a measurement."""
    import gnerf_harness
    dev = torch.device('cpu')
    if world > 1:
        import torch.distributed as dist
    x = torch.zeros(64)
    for _ in range(args.warmup):
        x += 1
    # rehearsal of one_solver_search: the stub "convolution" finds its solver in this rank's MIOpen user directory or "searches" for
    # GNERF_BENCH_STUB_SEARCH_S seconds and records it there; with the shared search only rank 0 pays, whatever the number of ranks
    search = {'hit_before_first_conv': None, 'searched_s': 0.0}

    def stub_conv():
        d = os.environ.get('MIOPEN_USER_DB_PATH')
        rec = os.path.join(d, 'stub_solver.ufdb.txt') if d else None
        hit = bool(rec and os.path.isfile(rec))
        if search['hit_before_first_conv'] is None:
            search['hit_before_first_conv'] = hit
        if not hit:
            t = float(os.environ.get('GNERF_BENCH_STUB_SEARCH_S', '0.5'))
            time.sleep(t)
            search['searched_s'] += t
            if rec:
                with open(rec, 'w') as f:
                    f.write(f'searched by rank {rank}\n')
    t_search = time.perf_counter()
    one_solver_search(rank, world, stub_conv, (dist.barrier if world > 1 else (lambda: None)))
    search['wall_s'] = time.perf_counter() - t_search
    if os.environ.get('GNERF_BENCH_STUB_FAIL_RANK') == str(rank):       # rehearsal of a rank that dies while its peers wait in a collective
        os._exit(3)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x += 1
    mine = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = gnerf_harness.max_over_ranks(time.perf_counter() - t0, dev)
    ranks = rank_identity(rank, world, dev)
    if world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, search)
    else:
        everyone = [search]
    for r_, s_ in zip(ranks, everyone):
        r_['solver_search'] = s_
    assert float(x[0]) == args.steps + args.warmup and elapsed >= mine
    if rank == 0:
        line = {'metric': 'stub (no rendering): launcher / rendezvous / reduction rehearsal', 'stub': True, 'value': None, 'unit': None,
                'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
                'ranks_seen': dist.get_world_size() if world > 1 else 1, 'backend': dist.get_backend() if world > 1 else None,
                'ranks': ranks, 'self_launched': os.environ.get('GNERF_BENCH_SELF_LAUNCHED') == '1'}
        os.write(result_fd, (json.dumps(line) + '\n').encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--reps', type=int, default=5, help='minimum repetitions of the timed K-step region (the median is reported; more are run until 0.3 s are covered)')
    ap.add_argument('--producer-layout', action='store_true', help='headline = the step on channels_last planes + max |planes| from the plane producer (no layout change in the step)')
    ap.add_argument('--nchw-input', action='store_true', help='(default since round 3; kept for old command lines)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the gen_videos frames/sec measurement')
    ap.add_argument('--no-backward', action='store_true', help='skip the renderer-backward timing behind roofline_backward')
    ap.add_argument('--full-line', action='store_true', help='with several ranks: every rank runs the whole line (forced arithmetics, in-kernel draws, '
                    'backward, backbone-plane step, both orbit flows); by default those run on rank 0 alone and the others go straight to the orbit')
    ap.add_argument('--stub-step', action='store_true', help=argparse.SUPPRESS)       # CPU rehearsal of the multi-rank plumbing (stub_main)
    ap.add_argument('--cpu-baseline-worker', type=int, default=0, help=argparse.SUPPRESS)      # child of cpu_baseline(): the all-cores figure
    args = ap.parse_args()

    if args.cpu_baseline_worker:
        print(json.dumps(_cpu_oracle_rays_per_s(args.cpu_baseline_worker, 1, 20.0, max_passes=1)))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # RCCL refuses two ranks on one device, so say it here, readably -- from the driver's topology files, not through torch / HIP:
        # this parent makes no GPU-runtime call at all (visible_gpus_no_hip)
        if not args.stub_step and os.environ.get('GNERF_DIST_BACKEND', 'nccl') == 'nccl':
            seen = visible_gpus_no_hip()
            if seen is not None and seen < args.gpus:
                sys.exit(f'bench.py: --gpus {args.gpus} but {seen} GPU(s) visible (RCCL needs one device per rank; '
                         'GNERF_DIST_BACKEND=gloo rehearses several ranks on one card)')
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    if args.stub_step:
        os.environ['GNERF_DIST_BACKEND'] = 'gloo'

    # stdout carries exactly ONE line, the JSON result: libraries that chat on fd 1 (RCCL's version banner at communicator
    # creation, gloo's connection notes, plugin status lines) are sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import gnerf_harness
    rank, world, local_rank = gnerf_harness.init_from_env()        # nccl (= RCCL) when WORLD_SIZE > 1
    if world > 1:
        import torch.distributed as dist
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if world > 1:
        per_rank_miopen_env(rank)       # (a no-op under self_launch, which exported them; this covers torch.distributed.run)
    if args.stub_step:
        return stub_main(args, rank, world, result_fd)
    # (modulo only matters for a rehearsal of several ranks on a one-GPU box with GNERF_DIST_BACKEND=gloo)
    dev = torch.device('cuda', local_rank % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)

    import gnerf_hip
    gnerf_hip.load()                    # raises if libgnerf_hip.so is missing: no fallback
    planes, dec, c2w, intr = _scene(dev, seed=1000 + rank)
    torch.manual_seed(rank)
    rays_per_call = N_ITEMS * RES * RES
    # HIP events around the render call of every EV_STRIDE-th step of a timed region (on the launch stream).  Every step was instrumented
    # at first: the two event records cost ~10 us per step (0.571 ms per step without them, 0.583 with, tools/bench_step_graph.py) --
    # the probe was 2 % of what it measured.  One step in five keeps >= 10 samples per region and a fifth of that cost.
    EV_STRIDE = 5 if args.steps >= 10 else 1
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    timed_steps = [i for i in range(args.steps) if i % EV_STRIDE == 0]

    # The planes as this repo's plane PRODUCER hands them over: the backbone's last step, gnerf_upsample2x_add_nhwc,
    # writes channels_last memory [N,H,W,96] = the interleaved plane layout the render kernels address in place, and max |planes| with it.
    planes_cl = planes.reshape(N_ITEMS, 96, PLANE, PLANE).permute(0, 2, 3, 1).contiguous()
    amax_cl = gnerf_hip.planes_absmax(planes_cl)
    headline_nchw = not args.producer_layout

    # (Measured and dropped: ray generation and the two draws on a second stream beside the 100 MB repack, joined in front of the render
    # call -- 0.581 -> 0.616 ms per step, two runs each way on one box: the cross-stream dependencies cost more (the render call's own
    # HIP-event time grows from 0.509 to 0.541 ms) than the 20 us of small launches they hide.)
    def step(i=None, mlp='auto', nchw_input=True, generated=False):
        """generated: the render kernel makes its rays from the cameras and its two uniform draws from the device generator's Philox
        stream itself (gnerf_render_params ABI 8; same values as make_rays + torch.rand bit for bit, generator advanced alike) --
        no ray launch, no draw launches, no ray / noise tensors."""
        if nchw_input:
            nhwc, amax = gnerf_hip.planes_to_nhwc(planes, with_absmax=True)   # max |planes| rides on the repack: it picks the decoder arithmetic
        else:
            nhwc, amax = planes_cl, amax_cl
        if generated:
            o = d = noise_c = noise_f = None
            extra = dict(cameras=(c2w, intr, RES), rng=gnerf_hip.torch_philox_plan(dev, N_ITEMS, RES * RES, S_COARSE, S_FINE))
        elif FUSED_PREP:
            # round 6: the rays and the two draws in ONE launch -- gnerf_make_rays_and_draws: make_rays' rays, torch.rand's values bit for bit
            # (tests/test_gpu_parity.py::test_make_rays_and_draws_equal_the_three_launches), the generator advanced as the two calls would
            o, d, noise_c, noise_f = gnerf_hip.make_rays_and_draws(c2w, intr, RES, S_COARSE, S_FINE)
            extra = {}
        else:
            o, d = gnerf_hip.make_rays(c2w, intr, RES)
            noise_c = torch.rand([N_ITEMS, RES * RES, S_COARSE, 1], device=dev)
            noise_f = torch.rand(N_ITEMS * RES * RES, S_FINE, device=dev)
            extra = {}
        if i is not None and i % EV_STRIDE == 0:
            ev[i][0].record()
        out = gnerf_hip.render_forward(nhwc, N_ITEMS, dec, o, d, noise_c, noise_f, depth_resolution=S_COARSE,
                                       depth_resolution_importance=S_FINE, ray_start=RAY_START, ray_end=RAY_END,
                                       box_warp=BOX_WARP, image_width=RES, planes_absmax=amax, mlp=mlp, **extra)
        if i is not None and i % EV_STRIDE == 0:
            ev[i][1].record()
        return out

    def barrier():
        if world > 1:
            dist.barrier()

    def timed_region(mlp='auto', nchw_input=True, n_steps=None, generated=False, solo=False):
        """EXACTLY n_steps (default args.steps) steps between barrier + synchronize on both sides; (max-over-ranks seconds, this rank's
        seconds, mean render ms by HIP events on this rank).  solo: this rank alone, no collective (the side measurements rank 0 makes
        while the other ranks of a multi-GPU run wait at the orbit's first barrier)."""
        n_steps = args.steps if n_steps is None else n_steps
        torch.cuda.synchronize()
        if not solo:
            barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_steps):
            out = step(i if n_steps == args.steps else None, mlp, nchw_input, generated)
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        if solo:
            elapsed = mine
        else:
            barrier()
            elapsed = gnerf_harness.max_over_ranks(time.perf_counter() - t0, dev)
        assert torch.isfinite(out[0]).all()
        kms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in timed_steps) / len(timed_steps) if n_steps == args.steps else None
        return elapsed, mine, kms      # events sit on the launch stream around the render call

    def progress(what):            # one line per phase on stderr: a watchdog that kills silent runs sees the bench alive
        if rank == 0:
            print(f'[bench {time.strftime("%H:%M:%S")}] {what}', file=sys.stderr, flush=True)

    progress('scene ready; warm-up')
    for _ in range(args.warmup):
        step(None, 'auto', headline_nchw)
    # The GPU's clocks take ~50 ms of load to settle (the first repetitions of a cold run measured 0.73, 0.75, 0.69 ms per step, every
    # later one 0.65): keep stepping, untimed, until 0.2 s have passed, so that the timed repetitions measure the settled state
    torch.cuda.synchronize()
    t_ramp, ramp_steps = time.perf_counter(), 0
    while time.perf_counter() - t_ramp < 0.2:
        for _ in range(10):
            step(None, 'auto', headline_nchw)
        torch.cuda.synchronize()
        ramp_steps += 10
    # Repetitions of the K-step region: at least --reps, and as many as cover 0.3 s of stepping (K = 20 steps are 12 ms: five of them
    # gave the driver's round-2 run a 19 % spread; the median of ~25 does not move).  Every repetition is exactly K steps.
    chrono = [timed_region(nchw_input=headline_nchw)]
    n_reps = max(args.reps, min(60, int(0.3 / max(chrono[0][0], 1e-4)) + 1))
    if world > 1:
        n_reps = int(gnerf_harness.max_over_ranks(float(n_reps), dev))          # same count on every rank (the regions hold barriers)
    chrono += [timed_region(nchw_input=headline_nchw) for _ in range(n_reps - 1)]
    regions = sorted(chrono)                                                     # by elapsed time
    elapsed, mine, kernel_ms = regions[len(regions) // 2]                        # the median repetition is the one reported
    assert gnerf_hip.last_mlp_choice(dev) == 'f16x3', 'config 2 is inside the f16 hi/lo range: the device-side choice must pick it'
    # one long region (>= 100 ms of steps, whatever --steps is) as a cross-check of the K-step median
    n_long = max(args.steps, int(0.1 / (elapsed / args.steps)) + 1)
    if world > 1:
        n_long = int(gnerf_harness.max_over_ranks(float(n_long), dev))
    long_elapsed = sorted(timed_region(nchw_input=headline_nchw, n_steps=n_long)[0] for _ in range(3))[1]
    # the other plane layout, and the render call with each shipped decoder arithmetic forced
    other = sorted(timed_region(nchw_input=not headline_nchw) for _ in range(5))[2]
    # the shader clock while the bench step runs (a one-wave sampler on a side stream: gnerf_clock_sample): the roofline's peak is
    # cycles per second, and the chip holds well under the data sheet's 2.4 GHz under this kernel's load
    clock_mhz = None
    if rank == 0:
        try:
            clock_mhz = gnerf_hip.clock_under_load(lambda: [step(None, 'auto', headline_nchw) for _ in range(8)], microseconds=3000.0, device=dev)
        except Exception as e:
            clock_mhz = None
            progress(f'clock sample failed: {type(e).__name__}: {e}')
    progress('headline regions done; in-kernel rays / draws, forced arithmetics')
    # With several ranks the side measurements below (nothing in them scales: they describe the kernel) run on rank 0 ALONE, without
    # collectives, while the others go on to the orbit and wait at its first barrier: eight copies of them -- and eight concurrent
    # MIOpen solver searches for a flow nobody reads at N > 1 -- are what could push the first multi-GPU run over its time limit.
    # --full-line restores "every rank runs everything" (then with barriers, as at N = 1).
    full = world == 1 or args.full_line
    extras_here, solo = full or rank == 0, not full
    side_ranks = world if full else 1
    # the same two steps with rays and draws made inside the render kernel (SURVEY 8d "in-kernel Philox (throughput run -- state which)")
    inkernel = {}
    kernel_ms_by_mlp = {'auto': kernel_ms}
    if extras_here:
        try:
            for name, nchw in (('nchw_input', True), ('producer_layout', False)):
                step(None, 'auto', nchw, True)
                reg = sorted(timed_region(nchw_input=nchw, generated=True, solo=solo) for _ in range(5))[2]
                inkernel[name] = {'ms_per_step': 1e3 * reg[0] / args.steps, 'value': rays_per_call * args.steps * side_ranks / reg[0], 'render_call_ms': reg[2],
                                  'ranks_measured': side_ranks}
        except Exception as e:
            inkernel = {'error': f'{type(e).__name__}: {e}'[:300]}
        for mlp in ('f16x3', 'f32'):
            step(None, mlp, headline_nchw)
            kernel_ms_by_mlp[mlp] = sorted(timed_region(mlp, headline_nchw, solo=solo)[2] for _ in range(3))[1]
    per_rank = [{'rank': rank, 'value': rays_per_call * args.steps / mine, 'render_call_ms': kernel_ms}]
    if world > 1:
        t = torch.tensor([rays_per_call * args.steps / mine, kernel_ms], device=dev, dtype=torch.float64)
        allk = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allk, t)
        per_rank = [{'rank': r, 'value': float(x[0]), 'render_call_ms': float(x[1])} for r, x in enumerate(allk)]

    ranks = rank_identity(rank, world, dev)

    progress('renderer backward')
    backward = None
    if not args.no_backward and extras_here:
        try:
            backward = backward_times(dev, planes_cl, dec, c2w, intr)
            backward['ratio_to_forward_kernel'] = backward['call_ms']['staged_scatter'] / kernel_ms
        except Exception as e:
            backward = {'error': f'{type(e).__name__}: {e}'[:300]}

    realistic = None
    secondary = None
    if not args.no_secondary:
        try:
            del planes, planes_cl
            torch.cuda.empty_cache()
            if extras_here:
                try:
                    from torch_utils import custom_ops
                    custom_ops.verbosity = 'none'
                    progress('config 2 on backbone-output planes')
                    realistic = realistic_planes_step(dev, c2w, intr, args.steps, rank)
                except Exception as e:
                    realistic = {'error': f'{type(e).__name__}: {e}'[:300]}
            progress('gen_videos orbit (config 4)' + (', two flows' if full else ', fast flow (the reference flow is an N = 1 / --full-line figure)'))
            secondary = gen_videos_secondary(rank, world, dev, flows=('fast', 'reference') if full else ('fast',))
        except Exception as e:                                           # never lose the headline line to the secondary metric
            secondary = {'metric': 'frames/sec gen_videos', 'value': None, 'error': f'{type(e).__name__}: {e}'[:300]}

    if rank == 0:
        total_rays = rays_per_call * args.steps * world
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.isfile(tpath):
            traffic = json.load(open(tpath)).get('render_kernel_hbm_bytes_per_launch')
        pmc, pmc_source = latest_pmc()
        roof = roofline(kernel_ms_by_mlp['auto'], rays_per_call, S_COARSE, S_FINE, PLANE, N_ITEMS, pmc, pmc_source, traffic)     # the kernel the headline runs
        roof['clock'] = {'mhz_while_the_step_runs': clock_mhz, 'mhz_the_peak_is_priced_at': CLOCK_HZ / 1e6,
                         'frac_at_the_measured_clock': roof['frac'] * (CLOCK_HZ / 1e6) / clock_mhz if clock_mhz else None,
                         'note': 'gnerf_clock_sample: one wave on a side stream reads s_memtime against the 100 MHz s_memrealtime for 3 ms of steps.  `frac` stays '
                                 'priced at 2.4 GHz; at the clock the chip actually holds under this load the same kernel covers that much more of the '
                                 'cycles it is given (profiles/r05_stamps.txt: 1.94-2.00 GHz inside the render kernel itself)'}
        roof['build_head'] = build_head()
        roof['render_call_ms'] = kernel_ms_by_mlp
        roof['render_call_ms_note'] = ('HIP events around the render call of every 5th step inside the timed regions: auto = what the headline runs (every '
                                       'workgroup evaluates the range bounds itself and runs the f16x3 body here; `frac` is computed on it), f16x3 / f32 = '
                                       'that arithmetic forced; each includes the depth-clamp epilogue')
        nchw_name = 'NCHW planes as BASELINE.md section 2 defines the input; the NCHW -> [3N,H,W,32] layout change (100 MB read + written) and max |planes| inside every step'
        cl_name = 'planes as gnerf_upsample2x_add_nhwc writes them (channels_last [N,H,W,96] + max |planes|), read in place: no layout change inside the step'
        other_step = {'ms_per_step': 1e3 * other[0] / args.steps, 'value': total_rays / other[0], 'planes': nchw_name if not headline_nchw else cl_name}
        line = {
            'metric': 'rays/sec at 128^2 neural render, 96 depth samples',
            'value': total_rays / elapsed, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            # proof of what ran: ranks the process group saw, its backend ('nccl' = RCCL; None for one process), every rank's device
            'ranks_seen': dist.get_world_size() if world > 1 else 1, 'backend': dist.get_backend() if world > 1 else None,
            'self_launched': os.environ.get('GNERF_BENCH_SELF_LAUNCHED') == '1', 'ranks': ranks,
            'dtype': 'f32 (MLP products as compensated f16 hi/lo splits on MFMA, fp32 accumulate; exact-fp32 MFMA when the device-side range '
                     'check says so)', 'data': 'synthetic',
            'repetitions': {'n': len(regions), 'reported': 'median', 'untimed_clock_ramp_steps': ramp_steps,
                            'ms_per_step_min_median_max': [1e3 * regions[0][0] / args.steps, 1e3 * elapsed / args.steps, 1e3 * regions[-1][0] / args.steps],
                            'ms_per_step_in_run_order': [round(1e3 * e / args.steps, 4) for e, _, _ in chrono],
                            'value_min': total_rays / regions[-1][0], 'value_max': total_rays / regions[0][0],
                            'spread_frac': (regions[-1][0] - regions[0][0]) / elapsed,
                            'interquartile_spread_frac': (regions[(3 * len(regions)) // 4][0] - regions[len(regions) // 4][0]) / elapsed,
                            'one_region_of_100ms': {'steps': n_long, 'ms_per_step': 1e3 * long_elapsed / n_long,
                                                    'value': rays_per_call * n_long * world / long_elapsed}},
            'config': {'workload': 'config 2: 128x128 rays x (48+48) samples, 3x32x256x256 fp32 tri-planes, batch 4 per GPU; '
                                   + ('NCHW planes, repacked in every step' if headline_nchw else 'channels_last planes read in place'),
                       'step': ('NCHW->NHWC repack with max|planes| + ' if headline_nchw else '') + ('make_rays and the 2 torch.rand draws in one launch (gnerf_make_rays_and_draws: torch.rand\'s values bit for bit, generator advanced alike)' if FUSED_PREP else 'make_rays + 2 torch.rand draws') + ' + fused render kernel (device-side decoder-arithmetic choice) + depth clamp',
                       'planes': nchw_name if headline_nchw else cl_name,
                       'rays_per_step_per_gpu': rays_per_call, 'parallelism': f'rays sharded over {world} GPU(s), no data-path collective',
                       'producer_layout_value' if headline_nchw else 'nchw_input_value': other_step['value'],
                       'producer_layout_ms_per_step' if headline_nchw else 'nchw_input_ms_per_step': other_step['ms_per_step']},
            'producer_layout_step' if headline_nchw else 'nchw_input_step': other_step,
            'inkernel_rng_step': dict(inkernel, note='the step with ray generation and both uniform draws inside the render kernel (cameras + torch\'s Philox '
                                      'stream at the generator\'s offset; outputs and generator state equal the explicit-noise step\'s bit for bit: '
                                      'tests/test_gpu_parity.py::test_render_with_inkernel_rays_and_draws); `value` above is the explicit-noise step'),
            'per_rank': per_rank,
            'roofline': roof,
        }
        if realistic is not None:
            line['realistic_planes_step'] = realistic
        if backward is not None:
            line['roofline_backward'] = backward
        line['secondary'] = secondary
        line['side_measurements'] = 'every rank' if full else 'rank 0 alone, without collectives (inkernel_rng_step, forced arithmetics, roofline_backward, realistic_planes_step; orbit: fast flow only)'
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the CPU oracle on this host's cores, in the same run at every N (north_star): after the process group is gone, so the other
        # ranks are not held in a collective while rank 0 computes on the CPU for half a minute
        line['cpu_baseline'] = None
        if not args.no_cpu_baseline:
            progress('CPU oracle baseline')
            try:
                line['cpu_baseline'] = cpu_baseline()
            except Exception as e:                                      # never lose the line to the baseline
                line['cpu_baseline'] = {'value': None, 'error': f'{type(e).__name__}: {e}'[:300]}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + '\n').encode())


if __name__ == '__main__':
    main()
