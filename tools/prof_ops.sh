#!/bin/bash
# rocprofv3 evidence for the custom-op kernels (run on the GPU box through gpurun):   bash tools/prof_ops.sh <tag>
# Pass 1: --kernel-trace --stats over tools/bench_ops.py and tools/bench_filtered_lrelu.py.  Passes 2..: one PMC group each
# (FETCH_SIZE / WRITE_SIZE for HBM bytes, SQ instruction and LDS counters).  Summary -> gpurun_out/<tag>_ops_profile.json
tag=${1:-ops}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out
for script in bench_ops bench_filtered_lrelu; do
  timeout 240 rocprofv3 --kernel-trace --stats -d $out/stats_$script -o run -- python3 $R/tools/$script.py > $out/stats_$script.log 2>&1
  pass() { name=$1; shift; timeout 240 rocprofv3 --pmc "$@" -d $out/${name}_$script -o run -- python3 $R/tools/$script.py > $out/${name}_$script.log 2>&1; }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass sq SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES
done
cd $R && python3 - "$out" "$tag" <<'PY' && rm -rf $out
import collections, glob, json, os, sqlite3, sys
d, tag = sys.argv[1], sys.argv[2]
KERNELS = ['bias_act_rows_kernel', 'bias_act_kernel', 'upfirdn_blur4_kernel', 'upfirdn_fir4_kernel', 'upfirdn_tile_kernel', 'upfirdn_generic_kernel',
           'filtered_lrelu_fused_kernel', 'lrelu_act_kernel', 'grid_sample_fwd_kernel', 'grid_sample_bwd_kernel', 'nchw_to_nhwc_kernel']
res = collections.defaultdict(dict)


def short(full, k):            # kernel name with its template arguments, without namespaces and parameter list
    i = full.find(k)
    j = full.find('(', i)
    return full[i:j if j > 0 else None].replace('(anonymous namespace)::', '')[:80]


for script in ('bench_ops', 'bench_filtered_lrelu'):
    f = glob.glob(os.path.join(d, 'stats_' + script, '**', '*.db'), recursive=True)
    if f:
        c = sqlite3.connect(f[0])
        for name, calls, tot, avg, pct in c.execute('select name, total_calls, total_duration, average, percentage from top_kernels'):
            for k in KERNELS:
                if k in name:
                    res[short(name, k)].update(calls=calls, avg_us=round(avg, 2))
    for grp in ('fetch', 'write', 'sq'):
        f = glob.glob(os.path.join(d, grp + '_' + script, '**', '*.db'), recursive=True)
        if not f:
            continue
        c = sqlite3.connect(f[0])
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for kname, disp, ctr, val in c.execute('select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name'):
            for k in KERNELS:
                if k in kname:
                    per[short(kname, k)][ctr].append(val)
        for key, ctrs in per.items():
            for ctr, vals in ctrs.items():
                res[key][ctr + '_mean'] = round(sum(vals) / len(vals), 1)
for key, r in res.items():
    if 'FETCH_SIZE_mean' in r and 'WRITE_SIZE_mean' in r:
        r['hbm_MB_mean_per_launch(FETCHx2+WRITE)'] = round((r['FETCH_SIZE_mean'] * 2 + r['WRITE_SIZE_mean']) * 1024 / 1e6, 2)
json.dump({'note': 'means over all launches of each kernel in tools/bench_ops.py + tools/bench_filtered_lrelu.py (mixed shapes / dtypes per kernel name); '
                   'FETCH_SIZE in KB, x2 for the gfx950 wide-load under-count (see traffic.json)', 'kernels': res},
          open(os.path.join('gpurun_out', f'{tag}_ops_profile.json'), 'w'), indent=1)
print(json.dumps(res, indent=1)[:5000])
PY
