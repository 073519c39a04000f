#!/bin/bash
# Round-6 experiment 28: non-temporal loads (1) / stores (2) / both (3) in the NCHW -> NHWC repack of the headline step, libraries alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp28
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
for v in base nt1 nt2 nt3 base nt1 nt2 nt3; do
  case $v in base) export GNERF_HIP_LIB="$V/libgnerf_base.so";; *) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_REPACK_NT=${v#nt}.so";; esac
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --reps 5 --no-cpu-baseline --no-secondary --no-backward 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'lib': '$v', 'value_Mrays': round(d['value'] / 1e6, 2), 'ms_per_step': round(d['ms_per_step'], 4), 'render_call_ms': round(d['roofline'].get('kernel_ms'), 4)}))" | tee -a $O/repack_nt.jsonl || exit 1
done
