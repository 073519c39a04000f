#!/bin/bash
# Round-6 experiment 3: the fp32-grade convolution (split + OUT32 kernel) -- kernel parity against float64, configs 3 / 5 against the fixtures with
# the route pinned, then config 3's time with the route on and off (GNERF_F32X3) and the convolution kernels' timing after the roll of the
# transposed form's phase loop.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp3
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/tests.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_f32x3 or conv3x3 or conv_transpose or config3 or config5 or fast_modconv or frozen_generator or inference_mode" 2>&1 | tail -25 | tee -a $O/tests.txt
grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
: > $O/config3.jsonl
for f in 1 0; do
  export GNERF_F32X3=$f
  echo "== GNERF_F32X3=$f" | tee -a $O/config3.jsonl
  timeout -k 10 400 python3 tools/bench_generator.py --only 3 2>/dev/null | grep '^{' | cut -c1-700 | tee -a $O/config3.jsonl
done
unset GNERF_F32X3
timeout -k 10 200 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('conv', d['shape'], 'ms', round(d['fused_ms'], 4), 'PFLOPs', round(d['fused_PFLOPs'], 3))" | tee $O/conv_timing.txt
timeout -k 10 200 python3 tools/bench_conv_transpose.py --search 0 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('transposed', d['shape'], 'ms', round(d['fused_ms'], 4), 'PFLOPs', round(d['PFLOPs'], 3))" | tee -a $O/conv_timing.txt
