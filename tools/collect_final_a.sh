#!/bin/bash
# Round-end evidence, part A (run on the GPU box through gpurun): rocprofv3 kernel stats + PMC passes of the bench command, the
# bench line itself, the backward kernels' atomic counters.  Summaries land in profiles/ (copied back through gpurun_out/profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
tag=${1:-r02_final}
bash tools/prof_forward.sh $tag > gpurun_out/prof_forward.log 2>&1
rm -rf gpurun_out/prof_$tag
python bench.py --steps 50 --warmup 5 --reps 5 > gpurun_out/profiles/${tag}_bench.json 2> gpurun_out/bench.err
tail -c 400 gpurun_out/profiles/${tag}_bench.json
bash tools/prof_bwd.sh > gpurun_out/prof_bwd.log 2>&1
cp gpurun_out/r02_backward_profile.json gpurun_out/profiles/ 2>/dev/null
ls -la gpurun_out/profiles
