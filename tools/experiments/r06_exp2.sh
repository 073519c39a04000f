#!/bin/bash
# Round-6 experiment 2: (a) the generator at a batch of ONE on the shared-weight form / the overlay's conv2d_resample on csrc/conv3x3.hip: the
# route-pinning tests, then the orbit frame by frame in both flows with the route on and off; (b) counters of the forward kernel, this tree
# against the round-5 library (what did the instruction diet remove, and what did the SIMDs do with it?).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp2
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/tests.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3 or fast_modconv or frozen_generator or orbit or views or inference_mode or per_latent or cache or conv3x3 or conv_transpose" 2>&1 | tail -8 | tee -a $O/tests.txt
grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
: > $O/orbit.jsonl
for mode in "fast 1 1" "fast 0 1" "fast 1 0" "ref 1 1" "ref 0 1"; do
  set -- $mode
  export GNERF_FUSED_CONV=$2 GNERF_SHARED_AT_ONE=$3
  flag=""; [ $1 = ref ] && flag="--reference-flow"
  echo "== $1 flow, GNERF_FUSED_CONV=$2 GNERF_SHARED_AT_ONE=$3" | tee -a $O/orbit.jsonl
  timeout -k 10 300 python3 tools/bench_generator.py --only 4 --frames 60 $flag 2>/dev/null | grep '^{' | cut -c1-600 | tee -a $O/orbit.jsonl
done
unset GNERF_FUSED_CONV GNERF_SHARED_AT_ONE
for v in default r05; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so"; fi
  bash tools/prof_insts.sh r06_forward_$v render_kernel_pipe tools/ablate.py > $O/insts_$v.txt 2>&1
  cp gpurun_out/r06_forward_${v}_insts.json $O/ 2>/dev/null
done
unset GNERF_HIP_LIB
tail -40 $O/insts_default.txt
# (c) gate for an fp32-grade mode of the convolution kernel: MIOpen's fp32 3x3 against the f16 kernel on [hi | lo | hi] x [hi | hi | lo] channels
timeout -k 10 300 python3 tools/bench_conv_f32grade.py --search 1 2>/dev/null | grep '^{' | tee $O/conv_f32grade.jsonl
