#!/bin/bash
# Round-6 experiment 29: the whole GPU suite on the final tree (+ non-temporal output stores in the plain convolution), then the round's evidence, part a (rocprofv3 kernel statistics and counter
# passes of the bench command, the bench line, the backward profile).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp29
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/suite.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -12 | tee -a $O/suite.txt
grep -q "failed\|error\|core dump" $O/suite.txt && { echo "stopping" | tee -a $O/suite.txt; exit 1; }
RND=r06 bash tools/collect_round.sh a r06_final 2>&1 | tail -30
