#!/bin/bash
# Round-5 experiment 11: the shared epilogue with paired fp16 rounding / scaling and the conv's clamp unswitched: GPU suite, conv and blur timings.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp11
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/suite.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee -a $O/suite.txt
echo "== conv SR shapes" | tee $O/conv.txt
timeout -k 10 300 python3 tools/bench_conv3x3.py --shapes sr --search 1 2>&1 | tail -2 | cut -c1-1500 | tee -a $O/conv.txt
echo "== ops" | tee $O/ops.txt
timeout -k 10 300 python3 tools/bench_ops.py 2>/dev/null | grep -i "blur\|epilogue" | cut -c1-300 | tee -a $O/ops.txt
