#!/bin/bash
# Round-5 experiment 15: is the forward kernel of the final tree the kernel of d945919?  (Two end-of-round collections on two boxes differ by
# 5.6 %.)  The library built from `git archive d945919` against the final one, alternating on ONE box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp15
mkdir -p $O
: > $O/ab.txt
for v in default d945919 default d945919 default d945919; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so"; fi
  timeout -k 10 120 python3 tools/ablate.py "$v" 2>/dev/null | tee -a $O/ab.txt
done
