#!/bin/bash
# Kernel statistics of the config-3 generator forward (batch 4) under rocprofv3 (run on the GPU box through gpurun):
#   -> gpurun_out/generator_kernel_stats.csv (Name, Calls, TotalDurationUs, AverageUs, Percentage)
# usage: bash tools/prof_generator.sh [bench_generator.py arguments; default --only 3]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g3
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g3 -o run -- python3 $R/tools/bench_generator.py ${@:---only 3} > /tmp/g3.log 2>&1
tail -2 /tmp/g3.log
cd $R && python3 - <<'PY'
import csv, glob, sqlite3
f = glob.glob('/tmp/prof_g3/**/*.db', recursive=True)[0]
rows = list(sqlite3.connect(f).execute('select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc'))
with open('gpurun_out/generator_kernel_stats.csv', 'w', newline='') as fh:
    w = csv.writer(fh)
    w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'Percentage'])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name[:200], calls, int(tot / 1000), round(avg / 1000, 1), round(pct, 4)])
for name, calls, tot, avg, pct in rows[:45]:
    print(f'{pct:5.1f}% {calls:5d} {avg / 1000:9.1f}  {name[:130]}')
PY
