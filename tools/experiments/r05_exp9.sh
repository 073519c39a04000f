#!/bin/bash
# Round-5 experiment 9: what the fused conv waits for -- timing-only builds without the per-step weight fetch / the later input chunks / the per-step barrier.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp9
mkdir -p $O
: > $O/ablate.txt
for v in default CONVW CONVX CONVBAR CONVW+CONVX+CONVBAR default; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so"; fi
  timeout -k 10 200 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['shape'], 'fused_ms', round(d['fused_ms'], 4), 'PFLOPs', round(d['fused_PFLOPs'], 3))
" | tee -a $O/ablate.txt
done
