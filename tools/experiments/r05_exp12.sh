#!/bin/bash
# Round-5 experiment 12: channels_last fp16 blur on v_dot2_f32_f16 (pairs of columns): parity of everything that blurs, then the op timings.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp12
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/parity.txt
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "upfirdn or blur or conv2d_resample or config3 or config5 or generator or fast_modconv or overlay or orbit or views" 2>&1 | tail -5 | tee -a $O/parity.txt
echo "== ops" | tee $O/ops.txt
timeout -k 10 300 python3 tools/bench_ops.py 2>/dev/null | grep -i "blur" | cut -c1-300 | tee -a $O/ops.txt
