#!/bin/bash
# Round-5 experiment 3: the binned plane-gradient scatter -- parity, timing against the sorted form, kernel times.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp3
mkdir -p $O
echo "== parity" | tee $O/parity.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward or bit_reproducible or training_step" 2>&1 | tail -30 | tee -a $O/parity.txt
echo "== timing (binned default, then sorted)" | tee $O/timing.txt
for shape in "4 128" "4 64"; do
  BWD_TORCH=0 timeout -k 10 200 python3 tools/bench_bwd.py $shape 2>/dev/null | tee -a $O/timing.txt
  GNERF_BWD_SCATTER=sorted BWD_TORCH=0 timeout -k 10 200 python3 tools/bench_bwd.py $shape 2>/dev/null | tee -a $O/timing.txt
done
echo "== kernels" | tee $O/kernels.txt
cd /tmp && export TMPDIR=/tmp
for shape in "4 128" "4 64"; do
  tag=$(echo $shape | tr ' ' '_')
  BWD_ONLY=staged BWD_TORCH=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p3_$tag -o run -- python3 $R/tools/bench_bwd.py $shape > /tmp/p3_$tag.log 2>&1
  python3 - /tmp/p3_$tag "$shape" <<'PY' | tee -a $O/kernels.txt
import glob, sqlite3, sys
f = glob.glob(sys.argv[1] + '/**/*.db', recursive=True)
print('shape', sys.argv[2])
for name, calls, avg in sqlite3.connect(f[0]).execute('select name, total_calls, average from top_kernels order by total_duration desc limit 12'):
    print(f'  {avg:10.1f} us x {calls:4d}  {name[:110]}')
PY
done
