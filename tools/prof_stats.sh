#!/bin/bash
# rocprofv3 kernel-trace summary of any python tool (run on the GPU box through gpurun):
#   bash tools/prof_stats.sh <tag> tools/bench_generator.py --only 3
# writes gpurun_out/<tag>_kernel_stats.csv (per-kernel calls / total / average, sorted by total time)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats -d $out/stats -o run -- python3 $script "$@" > $out/stats.log 2>&1
cd $R && python3 - "$out" "$tag" <<'PY'
import glob, os, sqlite3, sys
d, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, 'stats', '**', '*.db'), recursive=True)
c = sqlite3.connect(f[0])
rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc').fetchall()
with open(os.path.join('gpurun_out', f'{tag}_kernel_stats.csv'), 'w') as fh:
    fh.write('Name,Calls,TotalDurationUs,AverageUs,Percentage\n')
    for name, calls, tot, avg, pct in rows:
        short = name if len(name) < 140 else name[:80] + '...' + name[-50:]
        fh.write('"%s",%d,%d,%.1f,%.4f\n' % (short.replace('"', "'"), calls, tot, avg, pct))
print(open(os.path.join('gpurun_out', f'{tag}_kernel_stats.csv')).read()[:6000])
PY
