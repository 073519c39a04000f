#!/bin/bash
# Round-5 experiment 7: clocks under different loads; counters of the fused conv; binned scatter with 8 x 8 plane tiles; orbit profile.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp7
mkdir -p $O
echo "== clocks" | tee $O/clock.txt
timeout -k 10 200 python3 tools/clock_check.py 2>/dev/null | tail -1 | tee -a $O/clock.txt
echo "== binned scatter: 16 x 16 (default) vs 8 x 8 plane tiles" | tee $O/bin_tile.txt
V="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_D:GNERF_BIN_TILE=8.so"
for lib in default "$V" default "$V"; do
  if [ "$lib" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$lib"; fi
  for shape in "4 128" "4 64"; do
    echo "$(basename $lib) $shape: $(BWD_TORCH=0 BWD_ONLY=staged timeout -k 10 200 python3 tools/bench_bwd.py $shape 2>/dev/null | tail -1)" | tee -a $O/bin_tile.txt
  done
done
echo "== parity of the 8 x 8 build" | tee -a $O/bin_tile.txt
GNERF_HIP_LIB="$V" timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "backward or reproducible" 2>&1 | tail -4 | tee -a $O/bin_tile.txt
unset GNERF_HIP_LIB
echo "== fused conv under the profiler" | tee $O/conv_prof.txt
timeout -k 10 900 bash tools/prof_kernel.sh r05_conv3x3 conv3x3_epilogue_kernel tools/bench_conv3x3.py --shapes sr --search 0 --reps 5 2>&1 | tail -80 | tee -a $O/conv_prof.txt
echo "== orbit, 4 views per call" | tee $O/orbit.txt
timeout -k 10 500 bash tools/prof_orbit.sh r05_orbit_fast_views4 --frames-per-call 4 2>&1 | tail -5 | tee -a $O/orbit.txt
