#!/bin/bash
# Per-frame kernel statistics of the gen_videos orbit with the warm-up (MIOpen solver search) excluded (run on the GPU box):
#   bash tools/prof_orbit.sh <tag> [orbit_marked.py arguments]  ->  gpurun_out/<tag>_kernel_stats.csv + gpurun_out/<tag>_summary.json
# MARKED_SCRIPT=tools/config3_marked.py: the same for config 3 / config 5's generator passes (any script that brackets its timed part with the marker kernel
# and prints a JSON line with "frames")
tag=$1; shift
script=${MARKED_SCRIPT:-tools/orbit_marked.py}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/prof_$tag -o run -- python3 $R/$script "$@" > /tmp/prof_$tag.log 2>&1
tail -1 /tmp/prof_$tag.log
cd $R && python3 - "$tag" /tmp/prof_$tag /tmp/prof_$tag.log <<'PY'
import collections, csv, glob, json, re, sqlite3, sys
tag, d, log = sys.argv[1:4]
db = glob.glob(d + '/**/*.db', recursive=True)[0]
rows = list(sqlite3.connect(db).execute('select name, start, end, grid_x from kernels order by start'))
marks = [i for i, r in enumerate(rows) if 'torch_rand_kernel' in r[0] and r[3] in (424242, 424448)]
assert len(marks) >= 2, f'markers not found ({len(marks)})'
lo, hi = marks[-2], marks[-1]
inside = rows[lo + 1:hi]
run = json.loads([ln for ln in open(log) if ln.startswith('{')][-1])
frames = run['frames']


def family(name):
    n = name
    if 'render_kernel' in n: return 'ours: fused renderer'
    if 'conv3x3_epilogue_kernel' in n or 'split_f16x3' in n: return 'ours: 3x3 convolution (+ epilogue; fp32-grade form with its hi/lo split) on MFMA'
    if 'clamp_depth' in n or 'make_rays' in n or 'to_uint8' in n or 'planes_absmax' in n or 'absmax_kernel' in n: return 'ours: renderer side kernels'
    if 'blur4' in n or 'upfirdn' in n: return 'ours: upfirdn2d / blur (+ epilogue)'
    if 'bias_act' in n: return 'ours: bias_act'
    if 'torgb' in n or 'modconv' in n or 'scale_channels' in n or 'modulate' in n or 'normalise_styles' in n or 'upsample2x' in n: return 'ours: modulated-convolution surroundings'
    if 'distribution_elementwise' in n: return 'torch: uniform draws'
    if any(k in n for k in ('Conv', 'conv', 'gemm', 'Gemm', 'Cijk', 'igemm', 'xdlops', 'ck::', 'naive_conv', 'SubTensorOpWithScalar', 'transpose', 'batched_transpose')): return 'MIOpen / rocBLAS: convolutions'
    if 'at::native' in n or 'elementwise' in n: return 'torch: elementwise / copies'
    return 'other'


per = collections.defaultdict(lambda: [0, 0])
fam = collections.defaultdict(lambda: [0, 0])
for name, s, e, _ in inside:
    per[name][0] += 1; per[name][1] += e - s
    f = family(name); fam[f][0] += 1; fam[f][1] += e - s
tot = sum(v[1] for v in per.values())
span = rows[hi][1] - rows[lo][2]
with open(f'gpurun_out/{tag}_kernel_stats.csv', 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['Name', 'Calls', 'CallsPerFrame', 'TotalDurationUs', 'AverageUs', 'UsPerFrame', 'Percentage'])
    for name, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        w.writerow([re.sub(r'\s+', ' ', name)[:200], c, round(c / frames, 2), round(t / 1e3, 1), round(t / c / 1e3, 2), round(t / frames / 1e3, 2), round(100 * t / tot, 3)])
summary = {'run': run, 'dispatches_between_markers': len(inside), 'dispatches_per_frame': len(inside) / frames,
           'gpu_busy_us_per_frame': tot / frames / 1e3, 'wall_us_per_frame_between_markers': span / frames / 1e3, 'gpu_busy_frac': tot / span,
           'families_us_per_frame': {k: {'us_per_frame': round(v[1] / frames / 1e3, 2), 'calls_per_frame': round(v[0] / frames, 2), 'pct_of_gpu_time': round(100 * v[1] / tot, 2)}
                                     for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])},
           'note': 'kernel dispatches between the two marker kernels of tools/orbit_marked.py only: the warm-up frames and MIOpen\'s solver search are not in these numbers'}
json.dump(summary, open(f'gpurun_out/{tag}_summary.json', 'w'), indent=1)
print(json.dumps(summary, indent=1)[:3000])
PY
