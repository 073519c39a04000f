#!/bin/bash
# Round-end parity evidence on the final build (run on the GPU box through gpurun): the whole GPU suite, then the long fuzz sweeps.
#   -> gpurun_out/${RND}_gpu_suite.txt, gpurun_out/${RND}_fuzz.jsonl (first line: build head)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export RND=${RND:-r05}
head=$(cat g-nerf_amd/gnerf_hip/BUILD_HEAD 2>/dev/null)
echo "build $head" > gpurun_out/${RND}_gpu_suite.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> gpurun_out/${RND}_gpu_suite.txt
tail -2 gpurun_out/${RND}_gpu_suite.txt
echo "{\"build_head\": \"$head\", \"GNERF_VERIFY_ABSMAX\": \"1\"}" > gpurun_out/${RND}_fuzz.jsonl
export GNERF_VERIFY_ABSMAX=1
timeout -k 10 1500 python3 tests/parity_tools/fuzz_render.py 1500 31 2> gpurun_out/fuzz_render.err >> gpurun_out/${RND}_fuzz.jsonl
tail -c 300 gpurun_out/fuzz_render.err
timeout -k 10 600 python3 tests/parity_tools/fuzz_backward.py 150 31 2> gpurun_out/fuzz_backward.err >> gpurun_out/${RND}_fuzz.jsonl
cut -c1-400 gpurun_out/${RND}_fuzz.jsonl
