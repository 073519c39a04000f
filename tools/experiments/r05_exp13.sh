#!/bin/bash
# Round-5 experiment 13: counters of the backward's kernels (tile kernel first: LDS conflicts, waits, MFMA busy).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
export BWD_ONLY=staged BWD_TORCH=0
timeout -k 10 900 bash tools/prof_kernel.sh r05_bwd_kernels "" tools/bench_bwd.py 4 128 2>&1 | tail -150
