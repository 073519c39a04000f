#!/bin/bash
# Round-5 experiment 5: fused conv v2 (software-pipelined k-steps): parity test, SR-shape timing, generator tests with the fused conv.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp5
mkdir -p $O
echo "== conv parity" | tee $O/parity.txt
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or fast_modconv or config3 or per_latent" 2>&1 | tail -25 | tee -a $O/parity.txt
echo "== conv SR shapes" | tee $O/conv.txt
timeout -k 10 500 python3 tools/bench_conv3x3.py --shapes sr --search 1 2>&1 | tail -2 | cut -c1-1500 | tee -a $O/conv.txt
