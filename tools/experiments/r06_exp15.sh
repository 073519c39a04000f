#!/bin/bash
# Round-6 experiment 15: the NCHW -> NHWC repack of the headline step with 16 / 32 dword loads in flight per lane (32 x 128 / 32 x 256 tiles) against
# the shipped 32 x 64 tile: the headline step of bench.py under each library, alternating, on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp15
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
for v in base px128 px256 base px128 px256; do
  case $v in base) unset GNERF_HIP_LIB;; px128) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_REPACK_PX=128.so";; px256) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_REPACK_PX=256.so";; esac
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --reps 5 --no-cpu-baseline --no-secondary --no-backward 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': '$v', 'value_Mrays': round(d['value'] / 1e6, 2), 'ms_per_step': round(d['ms_per_step'], 4), 'render_call_ms': d['roofline'].get('kernel_ms'), 'producer_layout_ms': d['producer_layout_step']['ms_per_step']}))" | tee -a $O/step_ab.jsonl || exit 1
done
