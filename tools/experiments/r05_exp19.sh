#!/bin/bash
# Round-5 experiment 19: several tiles per workgroup in the convolution kernel (a workgroup launch costs ~3.6 us): parity, then 1 / 2 / 4 / 8 tiles.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp19
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/parity.txt
for k in 3 default; do
  if [ $k = default ]; then unset GNERF_CONV_TILES_PER_WG; else export GNERF_CONV_TILES_PER_WG=$k; fi
  timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_transpose or conv3x3" 2>&1 | tail -2 | tee -a $O/parity.txt
  grep -q "failed\|error\|core dump" $O/parity.txt && { echo "stopping" | tee -a $O/parity.txt; exit 1; }
done
: > $O/k.txt
for k in 1 2 4 8 default; do
  if [ $k = default ]; then unset GNERF_CONV_TILES_PER_WG; else export GNERF_CONV_TILES_PER_WG=$k; fi
  echo "tiles per workgroup: $k" | tee -a $O/k.txt
  timeout -k 10 200 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  conv', d['shape'], 'ms', round(d['fused_ms'], 4), 'PFLOPs', round(d['fused_PFLOPs'], 3))
" | tee -a $O/k.txt
  timeout -k 10 200 python3 tools/bench_conv_transpose.py --search 0 2>/dev/null | tail -2 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('  transposed', d['shape'], 'ms', round(d['fused_ms'], 4), 'PFLOPs', round(d['PFLOPs'], 3))
" | tee -a $O/k.txt
done
