#!/bin/bash
# Round-6 experiment 12: the eight-wave convolution workgroup in both wave layouts (one row x 128 channels; two rows x 64 channels) against the
# shipped four-wave pipeline: parity tests under each variant, timings alternating between the three libraries, and the in-kernel clock of the
# channel-split layout.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp12
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/tests.txt
for v in "D:GNERF_CONV_WAVES=8+D:GNERF_CONV_COSPLIT=2" "D:GNERF_CONV_WAVES=8"; do
  GNERF_HIP_LIB=$V/libgnerf_$v.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or conv_transpose or conv_f32x3" 2>&1 | tail -3 | tee -a $O/tests.txt
  grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
done
for v in base w8 w8c2 base w8 w8c2; do
  case $v in base) unset GNERF_HIP_LIB;; w8) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_CONV_WAVES=8.so";; w8c2) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_CONV_WAVES=8+D:GNERF_CONV_COSPLIT=2.so";; esac
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
