#!/bin/bash
# Round-6 experiment 5: the pipelined forward kernel with its step barriers replaced by role-to-role progress counters in LDS (GNERF_PIPE_FLAGS=1):
# parity of the variant library first (under a short timeout: a counter that is never reached would hang the kernel), then A/B on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp5
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_PIPE_FLAGS=1.so"
echo "variant $GNERF_HIP_LIB" | tee $O/tests.txt
timeout -k 10 120 python3 tools/ablate.py flags_smoke 2>&1 | tail -1 | tee -a $O/tests.txt
grep -q '"ms"' $O/tests.txt || { echo "the variant did not finish a launch: stopping" | tee -a $O/tests.txt; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "render_golden or render_vs_oracle or instantiations_agree or full_size_properties or views_of_one or inkernel_rays or zero_weight or ragged" 2>&1 | tail -6 | tee -a $O/tests.txt
grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
timeout -k 10 300 python3 tools/determinism.py 2>&1 | tail -3 | tee -a $O/tests.txt
unset GNERF_HIP_LIB
: > $O/forward_ab.jsonl
for rep in 1 2 3; do
for v in default 'D:GNERF_PIPE_FLAGS=1'; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$V/libgnerf_$v.so"; fi
  timeout -k 10 120 python3 tools/ablate.py "$v" 2>/dev/null | tail -1 | tee -a $O/forward_ab.jsonl
done
done
