#!/bin/bash
# The backward's fuzz sweep alone, and smoke(), on the final build (the forward's 1 500-case sweep takes a whole gpurun call on a slow box).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export RND=${RND:-r05} GNERF_VERIFY_ABSMAX=1
head=$(cat g-nerf_amd/gnerf_hip/BUILD_HEAD 2>/dev/null)
echo "{\"build_head\": \"$head\", \"GNERF_VERIFY_ABSMAX\": \"1\", \"what\": \"fuzz_backward.py 150 31, then __graft_entry__.smoke()\"}" > gpurun_out/${RND}_fuzz_backward.jsonl
timeout -k 10 600 python3 tests/parity_tools/fuzz_backward.py 150 31 2> gpurun_out/fuzz_backward.err >> gpurun_out/${RND}_fuzz_backward.jsonl
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('{\"smoke\": \"ok\"}')" 2> gpurun_out/smoke.err | tail -1 >> gpurun_out/${RND}_fuzz_backward.jsonl
cut -c1-300 gpurun_out/${RND}_fuzz_backward.jsonl
