#!/bin/bash
# Round-5 experiment 17: the count walk takes the tile bounds the tile kernel records (no row reads): parity first, then timing.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp17
mkdir -p $O
set -o pipefail
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/parity.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reproducible" 2>&1 | tail -4 | tee -a $O/parity.txt
grep -q "passed" $O/parity.txt && ! grep -q "failed\|core dump\|error" $O/parity.txt || { echo "stopping: the first test did not pass" | tee -a $O/parity.txt; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward" 2>&1 | tail -4 | tee -a $O/parity.txt
grep -q "failed\|core dump\|error" $O/parity.txt && { echo "stopping: a backward test failed" | tee -a $O/parity.txt; exit 1; }
: > $O/ab.txt
for shape in "4 128" "4 64" "4 128"; do
  line="$(BWD_TORCH=0 BWD_ONLY=staged timeout -k 10 200 python3 tools/bench_bwd.py $shape 2>&1 | tail -1)"
  echo "default $shape: $line" | tee -a $O/ab.txt
  case "$line" in *bwd_ms*) ;; *) echo "stopping: the timing run failed" | tee -a $O/ab.txt; exit 1;; esac
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/p14 -o run -- python3 $R/tools/bench_bwd.py 4 128 > /tmp/p14.log 2>&1
python3 - <<'PY' | tee $O/kernels.txt
import glob, sqlite3
f = glob.glob('/tmp/p14/**/*.db', recursive=True)
if f:
    for name, calls, avg in sqlite3.connect(f[0]).execute('select name, total_calls, average from top_kernels'):
        if 'bin_' in name or 'render_bwd' in name or 'pipe_bwd' in name: print(round(avg, 1), 'us x', calls, name[:90])
PY
