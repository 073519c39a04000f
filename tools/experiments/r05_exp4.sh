#!/bin/bash
# Round-5 experiment 4: binned scatter v2 (scalar-load records) + the fused conv's first light.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp4
mkdir -p $O
echo "== conv small shapes" | tee $O/conv.txt
timeout -k 10 300 python3 tools/bench_conv3x3.py --shapes small --search 0 --reps 3 2>&1 | tail -8 | cut -c1-1500 | tee -a $O/conv.txt
echo "== conv SR shapes" | tee -a $O/conv.txt
timeout -k 10 500 python3 tools/bench_conv3x3.py --shapes sr --search 1 2>&1 | tail -4 | cut -c1-1500 | tee -a $O/conv.txt
bash tools/r05_exp3.sh
