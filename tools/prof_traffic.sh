#!/bin/bash
# HBM traffic of one python tool's kernels (run on the GPU box through gpurun): kernel stats + one FETCH_SIZE and one WRITE_SIZE pass
# (tools/prof_kernel.sh without its three SQ / TCC passes):
#   bash tools/prof_traffic.sh <tag> <kernel-name-substring> tools/<script>.py [args]  -> gpurun_out/<tag>_traffic.json
tag=$1; needle=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out
timeout 240 rocprofv3 --kernel-trace --stats -d $out/stats -o run -- python3 $script "$@" > $out/stats.log 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o run -- python3 $script "$@" > $out/fetch.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $out/write -o run -- python3 $script "$@" > $out/write.log 2>&1
cd $R && python3 - "$out" "$tag" "$needle" <<'PY'
import collections, glob, json, os, sqlite3, sys
d, tag, needle = sys.argv[1:4]
res = collections.defaultdict(dict)
f = glob.glob(os.path.join(d, 'stats', '**', '*.db'), recursive=True)
if f:
    for name, calls, tot, avg, pct in sqlite3.connect(f[0]).execute('select name, total_calls, total_duration, average, percentage from top_kernels'):
        if needle in name:
            res[name[:120]].update(calls=calls, avg_us=round(avg / 1000.0, 2) if avg > 1e5 else round(avg, 2))
for grp in ('fetch', 'write'):
    f = glob.glob(os.path.join(d, grp, '**', '*.db'), recursive=True)
    if not f:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for kname, disp, ctr, val in sqlite3.connect(f[0]).execute('select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name'):
        if needle in kname:
            per[kname[:120]][ctr].append(val)
    for key, ctrs in per.items():
        for ctr, vals in ctrs.items():
            res[key][ctr] = round(sum(vals) / len(vals), 1)
for key, r in res.items():
    if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
        r['hbm_MB_per_launch(FETCHx2+WRITE)'] = round((r['FETCH_SIZE'] * 2 + r['WRITE_SIZE']) * 1024 / 1e6, 2)     # as tools/prof_kernel.sh (the guide's gfx950 correction)
json.dump(res, open(os.path.join('gpurun_out', f'{tag}_traffic.json'), 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
