#!/bin/bash
# Round-5 experiment 2: slot reads in one round trip (new base) vs the round-4 kernel, with the pinned lookup orders / LDS layouts.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp2
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
C4="D:GNERF_TAP_STRIDE=28+D:GNERF_STAGE_SWZ=1+D:GNERF_LOOKUP_ROLL=4"; C1="D:GNERF_TAP_STRIDE=28+D:GNERF_STAGE_SWZ=1+D:GNERF_LOOKUP_ROLL=1"
echo "== ablate A/B" | tee $O/ab.txt
for v in r4base base "$C4" "$C1" r4base base "$C4" "$C1"; do
  GNERF_HIP_LIB="$V/libgnerf_$v.so" timeout -k 10 120 python3 tools/ablate.py "$v" 2>/dev/null | tee -a $O/ab.txt
done
echo "== stamps" | tee $O/stamps.txt
GNERF_HIP_LIB="$V/libgnerf_STAMPS.so" timeout -k 10 120 python3 tools/stamps.py 2>&1 | tail -40 | tee -a $O/stamps.txt
echo "== parity"
for v in base "$C4" "$C1"; do
  echo "-- $v" | tee -a $O/parity.txt
  GNERF_HIP_LIB="$V/libgnerf_$v.so" timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_stage or instantiations_agree or render_vs_oracle or views_of_one_item or backward_vs_oracle" 2>&1 | tail -25 | tee -a $O/parity.txt
done
