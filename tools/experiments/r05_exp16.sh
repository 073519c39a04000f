#!/bin/bash
# Round-5 experiment 16: the x2 layer's transposed convolution on csrc/conv3x3.hip (MODE 1): parity, timing against MIOpen, the generator's tests, the orbit.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp16
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/parity.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_transpose" 2>&1 | tail -4 | tee -a $O/parity.txt
grep -q "1 passed" $O/parity.txt || { echo "stopping: the kernel's own test did not pass" | tee -a $O/parity.txt; exit 1; }
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fast_modconv or config3 or config5 or orbit or views or generator or overlay or inference_mode or latent" 2>&1 | tail -4 | tee -a $O/parity.txt
timeout -k 10 300 python3 tools/bench_conv_transpose.py 2>&1 | tail -2 | cut -c1-400 | tee $O/timing.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3" 2>&1 | tail -3 | tee -a $O/parity.txt
timeout -k 10 500 bash tools/prof_orbit.sh r05_orbit_fast_views8 --frames-per-call 8 2>&1 | tail -3 | tee $O/orbit.txt
