#!/bin/bash
# Round-5 experiment 6: fused conv as two workgroups per CU (64-channel chunks, 76 KB of LDS each); then the whole GPU suite on this build.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp6
mkdir -p $O
echo "== conv parity" | tee $O/parity.txt
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or fast_modconv" 2>&1 | tail -25 | tee -a $O/parity.txt
echo "== conv SR shapes" | tee $O/conv.txt
timeout -k 10 500 python3 tools/bench_conv3x3.py --shapes sr --search 1 2>&1 | tail -2 | cut -c1-1500 | tee -a $O/conv.txt
echo "== GPU suite" | tee $O/suite.txt
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee -a $O/suite.txt
