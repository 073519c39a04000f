#!/bin/bash
# Round-5 experiment 5: fused conv v2 (software-pipelined k-steps): parity test, SR-shape timing, generator tests with the fused conv.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp5
mkdir -p $O
echo "== conv parity" | tee $O/parity.txt
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or fast_modconv or config3 or per_latent" 2>&1 | tail -25 | tee -a $O/parity.txt
echo "== conv SR shapes" | tee $O/conv.txt
timeout -k 10 500 python3 tools/bench_conv3x3.py --shapes sr --search 1 2>&1 | tail -2 | cut -c1-1500 | tee -a $O/conv.txt
echo "== one-wave-per-ray backward kernel: 2 waves per SIMD (spills) vs 1 (none)" | tee $O/bwd_wave.txt
for lib in default "D:GNERF_BWD_WAVE_OCC=1"; do
  if [ "$lib" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$lib.so"; fi
  for shape in "4 64" "2 32 160"; do
    echo "$lib $shape: $(GNERF_BWD_KERNEL=wave BWD_TORCH=0 BWD_ONLY=staged timeout -k 10 200 python3 tools/bench_bwd.py $shape 2>/dev/null | tail -1)" | tee -a $O/bwd_wave.txt
  done
done
unset GNERF_HIP_LIB
echo "== forward: no texel loads (what a perfect load latency could return)" | tee $O/ablate.txt
for v in default GATHER default GATHER; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so"; fi
  timeout -k 10 120 python3 tools/ablate.py "$v" 2>/dev/null | tee -a $O/ablate.txt
done
unset GNERF_HIP_LIB
