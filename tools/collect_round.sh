#!/bin/bash
# Round-end evidence (run on the GPU box through gpurun, one part per call to stay inside the call limit):
#   bash tools/collect_round.sh a [tag]   rocprofv3 kernel stats + PMC passes of the bench command, the bench line itself, the backward profile
#   bash tools/collect_round.sh b         per-op GB/s, backward times, training step, generator (configs 3 / 4), shapes, orbit, host overhead
#   bash tools/collect_round.sh c         generator kernel statistics (fast and reference flows), layer / layout micro-benchmarks
# Results land in gpurun_out/profiles/ and gpurun_out/$RND/ (copied into profiles/ afterwards).  RND defaults to r05.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
export RND=${RND:-r06}
part=${1:-a}
mkdir -p gpurun_out/profiles gpurun_out/$RND
case $part in
a)
  tag=${2:-${RND}_final}
  bash tools/prof_forward.sh $tag > gpurun_out/prof_forward.log 2>&1
  rm -rf gpurun_out/prof_$tag
  python bench.py --steps 50 --warmup 5 --reps 5 > gpurun_out/profiles/${tag}_bench.json 2> gpurun_out/bench.err
  tail -c 300 gpurun_out/profiles/${tag}_bench.json
  bash tools/prof_bwd.sh > gpurun_out/prof_bwd.log 2>&1
  cp gpurun_out/${RND}_backward_profile.json gpurun_out/profiles/ 2>/dev/null
  ls -la gpurun_out/profiles ;;
b)
  e=gpurun_out/$RND/ops.err; : > $e
  python tools/bench_ops.py > gpurun_out/$RND/ops_GBs.jsonl 2>> $e
  python tools/bench_bwd.py 4 64 > gpurun_out/$RND/backward_ms.jsonl 2>> $e
  python tools/bench_bwd.py 4 128 >> gpurun_out/$RND/backward_ms.jsonl 2>> $e
  python g-nerf_amd/train_step_mi355x.py --steps 10 > gpurun_out/$RND/train_step.jsonl 2>> $e
  python g-nerf_amd/train_step_mi355x.py --mode renderer --steps 20 >> gpurun_out/$RND/train_step.jsonl 2>> $e
  python tools/bench_generator.py > gpurun_out/$RND/generator.jsonl 2>> $e
  python tools/bench_generator.py --reference-flow >> gpurun_out/$RND/generator.jsonl 2>> $e
  python tools/bench_shapes.py > gpurun_out/$RND/shapes.jsonl 2>> $e
  python tools/bench_orbit.py > gpurun_out/$RND/orbit.jsonl 2>> $e
  python tools/bench_host_overhead.py > gpurun_out/$RND/host_overhead.jsonl 2>> $e
  tail -n 3 gpurun_out/$RND/*.jsonl | cut -c1-300
  tail -5 $e ;;
c)
  bash tools/prof_generator.sh --only 3 > gpurun_out/prof_generator.log 2>&1; cp gpurun_out/generator_kernel_stats.csv gpurun_out/$RND/generator_kernel_stats.csv
  # the orbit with the warm-up (MIOpen's solver search) excluded: tools/prof_orbit.sh -> gpurun_out/${RND}_orbit_*_{kernel_stats.csv,summary.json}
  bash tools/prof_orbit.sh ${RND}_orbit_fast_eager > gpurun_out/prof_orbit1.log 2>&1
  bash tools/prof_orbit.sh ${RND}_orbit_fast_views4 --frames-per-call 4 > gpurun_out/prof_orbit2.log 2>&1
  bash tools/prof_orbit.sh ${RND}_orbit_fast_graph --graph > gpurun_out/prof_orbit3.log 2>&1
  bash tools/prof_orbit.sh ${RND}_orbit_reference_graph --flow reference --graph > gpurun_out/prof_orbit4.log 2>&1
  head -8 gpurun_out/${RND}_orbit_*kernel_stats.csv | cut -c1-160 ;;
esac
