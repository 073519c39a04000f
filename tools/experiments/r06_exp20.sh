#!/bin/bash
# Round-6 experiment 20: the render kernel with the scalar wave's per-ray passes removed entirely (timing-only build PIPESCALAR: fine depths = coarse
# depths, constant weights) and without its rank merge (PIPERANK), against the shipped kernel, alternating on one box: the bound on what ANY
# reformulation of the scalar pass (the verdict's lane-per-ray form included) could return.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp20
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
for v in base PIPERANK PIPESCALAR base PIPERANK PIPESCALAR base PIPESCALAR; do
  GNERF_HIP_LIB=$V/libgnerf_$v.so timeout -k 10 200 python3 tools/ablate.py $v 2>/dev/null | grep '^{' | tee -a $O/scalar_wave_ablation.jsonl || exit 1
done
