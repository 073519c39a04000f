#!/bin/bash
# Round-6 experiment 19: the backward's fuzz sweep, smoke(), the determinism check and the packed-fp32 hazard probe on the final library.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
RND=r06 bash tools/collect_fuzz_backward.sh
timeout -k 10 300 python3 tools/determinism.py 2>/dev/null | tail -3 | tee gpurun_out/r06_determinism.txt
timeout -k 10 300 bash tools/probes/pk_opsel_hazard_probe.sh 2>&1 | tail -25 | tee gpurun_out/r06_hazard_probe.txt
