#!/bin/bash
# Round-6 experiment 1: the forward kernel's instruction diet (in-place DPP scans, lean tap set-up, role rotation, colour-bias table, per-plane ray
# words) and the convolution's fp32 epilogue: whole GPU suite first, then A/B against the round-5 library and one-change-off variants on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp1
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/suite.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee -a $O/suite.txt
grep -q "failed\|error\|core dump" $O/suite.txt && { echo "stopping" | tee -a $O/suite.txt; exit 1; }
V=$R/g-nerf_amd/gnerf_hip/variants
: > $O/forward_ab.jsonl
for rep in 1 2; do
for v in default r05 'D:GNERF_DPP_INPLACE=0' 'D:GNERF_TAPS_LEAN=0' 'D:GNERF_PIPE_ROTATE=0'; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$V/libgnerf_$v.so"; fi
  timeout -k 10 120 python3 tools/ablate.py "$v" 2>/dev/null | tail -1 | tee -a $O/forward_ab.jsonl
done
done
: > $O/conv_ab.txt
for v in default 'D:GNERF_CONV_EPILOGUE_F32=0' r05; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$V/libgnerf_$v.so"; fi
  echo "== $v" | tee -a $O/conv_ab.txt
  timeout -k 10 200 python3 tools/bench_conv3x3.py --shapes sr --search 0 2>/dev/null | grep '^{' | tee -a $O/conv_ab.txt
done
unset GNERF_HIP_LIB
