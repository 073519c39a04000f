#!/bin/bash
# Round-6 experiment 13: counted LDS waits in the eight-wave convolution loop (a pair of channel blocks starts when its own fragments have arrived)
# against one wait for all fragments; both in the two-rows-x-64-channels layout.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp13
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
A="$V/libgnerf_D:GNERF_CONV_WAVES=8+D:GNERF_CONV_COSPLIT=2.so"
B="$V/libgnerf_D:GNERF_CONV_WAVES=8+D:GNERF_CONV_COSPLIT=2+D:GNERF_CONV_COUNTED_WAITS=1.so"
GNERF_HIP_LIB=$B timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or conv_transpose or conv_f32x3" 2>&1 | tail -3 | tee $O/tests.txt
grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
for v in one counted one counted; do
  if [ $v = one ]; then export GNERF_HIP_LIB=$A; else export GNERF_HIP_LIB=$B; fi
  echo "== $v" | tee -a $O/conv3x3.jsonl $O/conv_transpose.jsonl $O/f32grade.jsonl
  timeout -k 10 300 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | grep '^{' | tee -a $O/conv3x3.jsonl | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['shape'], round(d['fused_ms'], 4))" || exit 1
  timeout -k 10 300 python3 tools/bench_conv_transpose.py --search 0 2>/dev/null | grep '^{' | tee -a $O/conv_transpose.jsonl | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    if 'fused_ms' in d: print('T', d['shape'], round(d['fused_ms'], 4))" || exit 1
  timeout -k 10 300 python3 tools/bench_conv_f32grade.py 2>/dev/null | grep '^{' | tee -a $O/f32grade.jsonl | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('F', d['shape'], d.get('own_f16x3_ms'))" || exit 1
done
