#!/bin/bash
# Round-6 experiment 17: whole-tile / phase-pair / single-phase jobs of the transposed convolution on the generator's shapes, then its parity test.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp17
mkdir -p $O
timeout -k 10 400 python3 tools/bench_convt_phase_jobs.py 2>/dev/null | grep '^{' | tee $O/convt_phase_jobs.jsonl
for m in 0 1 2; do GNERF_CONVT_PHASE_JOBS=$m timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv_transpose or conv_f32x3" 2>&1 | tail -2 | tee -a $O/tests.txt; done
