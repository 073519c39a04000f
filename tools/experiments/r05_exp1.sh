#!/bin/bash
# Round-5 experiment 1 (GPU box): A/B of the shade tile's lookup order and LDS layouts, the clock / placement stamps, and
# the counter attribution of LDS bank conflicts and waits.   bash tools/r05_exp1.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp1
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
S28="D:GNERF_TAP_STRIDE=28"; SWZ="D:GNERF_STAGE_SWZ=1"
echo "== ablate A/B" | tee $O/ab.txt
for v in base "$S28" "$SWZ" "$S28+$SWZ" "D:GNERF_LOOKUP_ROLL=4" "D:GNERF_LOOKUP_ROLL=1" "$S28+$SWZ+D:GNERF_LOOKUP_ROLL=4" "$S28+$SWZ+D:GNERF_LOOKUP_ROLL=1" base; do
  GNERF_HIP_LIB="$V/libgnerf_$v.so" timeout -k 10 120 python3 tools/ablate.py "$v" 2>/dev/null | tee -a $O/ab.txt
done
echo "== stamps" | tee $O/stamps.txt
GNERF_HIP_LIB="$V/libgnerf_STAMPS.so" timeout -k 10 120 python3 tools/stamps.py 2>&1 | tail -40 | tee -a $O/stamps.txt
echo "== parity of the pinned variants"
for v in "$S28+$SWZ+D:GNERF_LOOKUP_ROLL=1" "$S28+$SWZ+D:GNERF_LOOKUP_ROLL=4"; do
  GNERF_HIP_LIB="$V/libgnerf_$v.so" timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_stage or instantiations_agree or render_vs_oracle or views_of_one_item" 2>&1 | tail -3 | tee -a $O/parity.txt
done
echo "== counters"
for v in base "$S28+$SWZ+D:GNERF_LOOKUP_ROLL=1" SHADE; do
  tag=r05e1_$(echo "$v" | tr -c 'A-Za-z0-9\n' '_')
  GNERF_HIP_LIB="$V/libgnerf_$v.so" bash tools/prof_insts.sh $tag render_kernel_pipe tools/ablate.py > $O/pmc_$tag.txt 2>&1
  tail -30 $O/pmc_$tag.txt
done
