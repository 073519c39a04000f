#!/bin/bash
# Round-6 experiment 7: the two convolution parity tests with their new full-size cases; convolution timings (plain with the error fields, transposed);
# config 5's G + D step kernel by kernel (marker-bracketed); the orbit's reference flow under the marker profile.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp7
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/tests.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or conv_transpose or conv_f32x3" 2>&1 | tail -6 | tee -a $O/tests.txt
grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
timeout -k 10 300 python3 tools/bench_conv3x3.py --shapes sr --search 1 2>/dev/null | grep '^{' | tee $O/conv3x3.jsonl | cut -c1-900
timeout -k 10 300 python3 tools/bench_conv_transpose.py --search 1 2>/dev/null | grep '^{' | tee $O/conv_transpose.jsonl | cut -c1-400
MARKED_SCRIPT=g-nerf_amd/train_step_mi355x.py bash tools/prof_orbit.sh r06_config5 --marked --steps 5 --warmup 2 > $O/config5.txt 2>&1
cp gpurun_out/r06_config5_kernel_stats.csv gpurun_out/r06_config5_summary.json $O/ 2>/dev/null
head -c 3000 $O/config5.txt
bash tools/prof_orbit.sh r06_orbit_reference_graph --flow reference --graph > $O/orbit_ref.txt 2>&1
cp gpurun_out/r06_orbit_reference_graph_kernel_stats.csv gpurun_out/r06_orbit_reference_graph_summary.json $O/ 2>/dev/null
head -c 2500 $O/orbit_ref.txt
