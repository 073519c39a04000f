#!/bin/bash
# Per-kernel rocprofv3 evidence for any python tool (run on the GPU box through gpurun):
#   bash tools/prof_kernel.sh <tag> <kernel-name-substring> tools/<script>.py [args]
# Pass 1 --kernel-trace --stats; then one --pmc pass per counter group (never combined with tracing).  Means per launch of the
# kernels whose name contains the substring -> gpurun_out/<tag>_kernel_profile.json
tag=$1; needle=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out
timeout 240 rocprofv3 --kernel-trace --stats -d $out/stats -o run -- python3 $script "$@" > $out/stats.log 2>&1
pass() { name=$1; shift; timeout 240 rocprofv3 --pmc "$@" -d $out/$name -o run -- python3 $script "${ARGS[@]}" > $out/$name.log 2>&1; }
ARGS=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
cd $R && python3 - "$out" "$tag" "$needle" <<'PY'
import collections, glob, json, os, sqlite3, sys
d, tag, needle = sys.argv[1:4]
res = collections.defaultdict(dict)
f = glob.glob(os.path.join(d, 'stats', '**', '*.db'), recursive=True)
if f:
    for name, calls, tot, avg, pct in sqlite3.connect(f[0]).execute('select name, total_calls, total_duration, average, percentage from top_kernels'):
        if needle in name:
            res[name[:160]].update(calls=calls, avg_us=round(avg / 1000.0, 2) if avg > 1e5 else round(avg, 2))
for grp in ('fetch', 'write', 'sq1', 'sq2', 'mfma', 'tcc'):
    f = glob.glob(os.path.join(d, grp, '**', '*.db'), recursive=True)
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for kname, disp, ctr, val in sqlite3.connect(f[0]).execute('select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name'):
        if needle in kname:
            per[kname[:160]][ctr].append(val)
    for key, ctrs in per.items():
        for ctr, vals in ctrs.items():
            res[key][ctr] = round(sum(vals) / len(vals), 1)
for key, r in res.items():
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
        r['hbm_MB_per_launch(FETCHx2+WRITE)'] = round((r['FETCH_SIZE'] * 2 + r['WRITE_SIZE']) * 1024 / 1e6, 2)
json.dump(res, open(os.path.join('gpurun_out', f'{tag}_kernel_profile.json'), 'w'), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
rm -rf $out            # the raw rocpd files are large: gpurun copies back at most 64 MiB
