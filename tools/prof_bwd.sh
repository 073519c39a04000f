#!/bin/bash
# Backward kernels at BASELINE config 2 under rocprofv3 (run on the GPU box through gpurun): kernel stats, then the atomic
# request counters in their own --pmc passes, for the staged (two-pass) and the single-pass forms.
#   -> gpurun_out/${RND:-r05}_backward_profile.json   (staged = render_kernel_pipe_bwd + render_bwd_tiles_kernel + the binned scatter's five
#      kernels (bin_walk x 2, bin_scan, bin_accumulate, bin_halo) on the pipelined path; `sorted` = the same call with
#      GNERF_BWD_SCATTER=sorted: round 2's plane_scatter_kernel as the second pass; `wave` = GNERF_BWD_KERNEL=wave: the
#      one-wave-per-ray kernel every other shape runs; `single` = the single-pass form)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp BWD_TORCH=0
out=$R/gpurun_out/prof_bwd
rm -rf $out; mkdir -p $out
for form in staged sorted wave single; do
  export BWD_ONLY=$form; unset GNERF_BWD_KERNEL GNERF_BWD_SCATTER
  if [ $form = wave ]; then export BWD_ONLY=staged GNERF_BWD_KERNEL=wave; fi
  if [ $form = sorted ]; then export BWD_ONLY=staged GNERF_BWD_SCATTER=sorted; fi
  timeout 200 rocprofv3 --kernel-trace --stats -d $out/${form}_stats -o run -- python3 $R/tools/bench_bwd.py 4 128 > $out/${form}_stats.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum -d $out/${form}_atomic -o run -- python3 $R/tools/bench_bwd.py 4 128 > $out/${form}_atomic.log 2>&1
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d $out/${form}_fetch -o run -- python3 $R/tools/bench_bwd.py 4 128 > $out/${form}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d $out/${form}_write -o run -- python3 $R/tools/bench_bwd.py 4 128 > $out/${form}_write.log 2>&1
done
cd $R && python3 - "$out" <<'PY'
import collections, glob, json, os, sqlite3, sys
d = sys.argv[1]
res = {}
for form in ('staged', 'sorted', 'wave', 'single'):
    r = collections.defaultdict(dict)
    f = glob.glob(os.path.join(d, form + '_stats', '**', '*.db'), recursive=True)
    if f:
        for name, calls, avg in sqlite3.connect(f[0]).execute('select name, total_calls, average from top_kernels'):
            if 'render_bwd' in name or 'plane_scatter_kernel' in name or 'render_kernel_pipe_bwd' in name or '::bin_' in name:
                r[name.split('(anonymous namespace)::')[1].split('(')[0]].update(calls=calls, avg_ms=round(avg / 1000.0, 3))      # top_kernels.average is in microseconds
    for grp in ('atomic', 'fetch', 'write'):
        f = glob.glob(os.path.join(d, f'{form}_{grp}', '**', '*.db'), recursive=True)
        if not f:
            r['_missing'][grp] = open(os.path.join(d, f'{form}_{grp}.log')).read()[-300:]
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for kname, disp, ctr, val in sqlite3.connect(f[0]).execute('select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name'):
            if 'render_bwd' in kname or 'plane_scatter_kernel' in kname or 'render_kernel_pipe_bwd' in kname or '::bin_' in kname:
                per[kname.split('(anonymous namespace)::')[1].split('(')[0]][ctr].append(val)
        for k, ctrs in per.items():
            for ctr, vals in ctrs.items():
                r[k][ctr] = round(sum(vals) / len(vals), 1)
    res[form] = r
try:
    res['head'] = open('g-nerf_amd/gnerf_hip/BUILD_HEAD').read().strip()
except OSError:
    res['head'] = None
json.dump(res, open("gpurun_out/" + os.environ.get("RND", "r05") + "_backward_profile.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out
