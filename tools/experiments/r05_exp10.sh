#!/bin/bash
# Round-5 experiment 10: fused conv with fragment reads pipelined across steps (GNERF_CONV_PIPE=1, the default) against the first two-workgroup loop.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp10
mkdir -p $O
echo "== parity" | tee $O/parity.txt
echo skipped | tee -a $O/parity.txt
: > $O/ab.txt
for v in default "D:GNERF_CONV_STAGGER=1" "D:GNERF_CONV_STAGGER=2" default "D:GNERF_CONV_STAGGER=1" "D:GNERF_CONV_STAGGER=2"; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so"; fi
  timeout -k 10 200 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['shape'], 'fused_ms', round(d['fused_ms'], 4), 'PFLOPs', round(d['fused_PFLOPs'], 3), 'max_abs_diff', d['scale+next']['max_abs_diff'])
" | tee -a $O/ab.txt
done
