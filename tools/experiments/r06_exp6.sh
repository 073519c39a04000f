#!/bin/bash
# Round-6 experiment 6: the whole GPU suite on the current tree, then the orbit (bench_generator config 4, frame by frame and batched), config 3.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp6
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/suite.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee -a $O/suite.txt
grep -q "failed\|error\|core dump" $O/suite.txt && { echo "stopping" | tee -a $O/suite.txt; exit 1; }
timeout -k 10 400 python3 tools/bench_generator.py --frames-per-call 8 2>/dev/null | grep '^{' | cut -c1-500 | tee $O/generator.jsonl
