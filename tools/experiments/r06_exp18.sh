#!/bin/bash
# Round-6 experiment 18: the forward's 1 500-case fuzz sweep on the final library (GNERF_VERIFY_ABSMAX=1).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
head=$(cat g-nerf_amd/gnerf_hip/BUILD_HEAD 2>/dev/null)
echo "{\"build_head\": \"$head\", \"GNERF_VERIFY_ABSMAX\": \"1\", \"what\": \"fuzz_render.py 1500 31\"}" > gpurun_out/r06_fuzz.jsonl
export GNERF_VERIFY_ABSMAX=1
timeout -k 10 1120 python3 tests/parity_tools/fuzz_render.py 1500 31 2> gpurun_out/fuzz_render.err >> gpurun_out/r06_fuzz.jsonl
echo "rc $?"
tail -c 300 gpurun_out/fuzz_render.err
tail -n 2 gpurun_out/r06_fuzz.jsonl | cut -c1-600
