#!/bin/bash
# Round-6 experiment 26: non-temporal input-tile loads (1), output stores (2) or both (3) in the convolution kernel against the shipped default
# cache policy: plain, transposed and fp32-grade timings, libraries alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp26
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
for v in base nt1 nt2 nt3 base nt1 nt2 nt3; do
  case $v in base) export GNERF_HIP_LIB="$V/libgnerf_base.so";; nt1) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_CONV_NT=1.so";; nt2) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_CONV_NT=2.so";; nt3) export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_CONV_NT=3.so";; esac
  echo "== $v" | tee -a $O/conv_nt.txt
  timeout -k 10 300 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['shape'], round(d['fused_ms'], 4))" | tee -a $O/conv_nt.txt || exit 1
  timeout -k 10 300 python3 tools/bench_conv_transpose.py --search 0 2>/dev/null | grep '^{' | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    if 'fused_ms' in d: print('T', d['shape'], round(d['fused_ms'], 4))" | tee -a $O/conv_nt.txt || exit 1
done
