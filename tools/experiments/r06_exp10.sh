#!/bin/bash
# Round-6 experiment 10: A/B of the convolution workgroup with EIGHT waves (GNERF_CONV_WAVES=8: a wave owns one tile row, 64 accumulator registers,
# <= 128 registers -> four waves per SIMD, fragments single-buffered) against the shipped four-wave pipeline: parity tests under the variant, then
# both libraries' timings on the SR shapes, the transposed form and the fp32-grade form.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp10
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants/libgnerf_D:GNERF_CONV_WAVES=8.so
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/tests.txt
GNERF_HIP_LIB=$V timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or conv_transpose or conv_f32x3" 2>&1 | tail -6 | tee -a $O/tests.txt
grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
for v in base w8 base w8; do
  if [ $v = w8 ]; then export GNERF_HIP_LIB=$V; else unset GNERF_HIP_LIB; fi
  echo "== $v" | tee -a $O/conv3x3.jsonl $O/conv_transpose.jsonl $O/f32grade.jsonl
  timeout -k 10 300 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | grep '^{' | tee -a $O/conv3x3.jsonl | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print({k: d[k] for k in d if k in ('shape', 'cin', 'cout', 'res', 'n', 'fused_ms', 'own_ms', 'ms', 'tflops', 'fused_tflops')})" || exit 1
  timeout -k 10 300 python3 tools/bench_conv_transpose.py --search 0 2>/dev/null | grep '^{' | tee -a $O/conv_transpose.jsonl | cut -c1-300 || exit 1
  timeout -k 10 300 python3 tools/bench_conv_f32grade.py 2>/dev/null | grep '^{' | tee -a $O/f32grade.jsonl | cut -c1-300 || exit 1
done
