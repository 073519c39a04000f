#!/bin/bash
# Round-end evidence, part B: per-op GB/s, backward times, training step, generator (configs 3 / 4) + its kernel statistics,
# shapes, orbit, host overhead.  -> gpurun_out/r02/*.jsonl (copied into profiles/r02_*.jsonl afterwards)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
bash tools/collect_r02.sh > gpurun_out/collect_r02.log 2>&1
python tools/bench_host_overhead.py > gpurun_out/r02/host_overhead.jsonl 2>> gpurun_out/r02/ops.err
GNERF_HIP_BINDING=ctypes python tools/bench_host_overhead.py >> gpurun_out/r02/host_overhead.jsonl 2>> gpurun_out/r02/ops.err
python tools/bench_layers.py > gpurun_out/r02/generator_layers.jsonl 2>> gpurun_out/r02/ops.err
python tools/bench_sr_conv_layout.py > gpurun_out/r02/sr_conv_layout.jsonl 2>> gpurun_out/r02/ops.err
python tools/bench_sr_conv_layout.py 1 >> gpurun_out/r02/sr_conv_layout.jsonl 2>> gpurun_out/r02/ops.err
bash tools/prof_generator.sh > gpurun_out/prof_generator.log 2>&1
tail -n 2 gpurun_out/r02/*.jsonl | cut -c1-400
tail -5 gpurun_out/r02/ops.err
