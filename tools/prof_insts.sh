#!/bin/bash
# Quick instruction-count A/B of one python tool's kernels (run on the GPU box through gpurun):
#   bash tools/prof_insts.sh <tag> <kernel-name-substring> tools/<script>.py [args]
# One --pmc pass (SQ instruction counters) + one --kernel-trace --stats pass -> gpurun_out/<tag>_insts.json
tag=$1; needle=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $out/stats -o run -- python3 $script "$@" > $out/stats.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $out/sq1 -o run -- python3 $script "$@" > $out/sq1.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES -d $out/sq2 -o run -- python3 $script "$@" > $out/sq2.log 2>&1
cd $R && python3 - "$out" "$tag" "$needle" <<'PY'
import collections, glob, json, os, sqlite3, sys
d, tag, needle = sys.argv[1:4]
res = collections.defaultdict(dict)
f = glob.glob(os.path.join(d, 'stats', '**', '*.db'), recursive=True)
if f:
    for name, calls, tot, avg, pct in sqlite3.connect(f[0]).execute('select name, total_calls, total_duration, average, percentage from top_kernels'):
        if needle in name:
            res[name[:100]].update(calls=calls, avg_us=round(avg / 1000.0, 2) if avg > 1e5 else round(avg, 2))
for f in [x for grp in ('sq1', 'sq2') for x in glob.glob(os.path.join(d, grp, '**', '*.db'), recursive=True)[:1]]:
    f = [f]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for kname, disp, ctr, val in sqlite3.connect(f[0]).execute('select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name'):
        if needle in kname:
            per[kname[:100]][ctr].append(val)
    for key, ctrs in per.items():
        for ctr, vals in ctrs.items():
            res[key][ctr] = round(sum(vals) / len(vals), 1)
json.dump(res, open(os.path.join('gpurun_out', f'{tag}_insts.json'), 'w'), indent=1)
print(json.dumps(res, indent=1)[:4000])
PY
