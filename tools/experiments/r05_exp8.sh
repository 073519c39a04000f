#!/bin/bash
# Round-5 experiment 8: conv with conflict-free input-tile swizzle, operands by LDS-DMA, launcher-checked lrelu slope; blur + epilogue with the same.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp8
mkdir -p $O
echo "== parity" | tee $O/parity.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "conv3x3 or fast_modconv or blur or upfirdn or modconv or config3 or epilogue" 2>&1 | tail -4 | tee -a $O/parity.txt
echo "== conv SR shapes" | tee $O/conv.txt
timeout -k 10 500 python3 tools/bench_conv3x3.py --shapes sr --search 1 2>&1 | tail -2 | cut -c1-1500 | tee -a $O/conv.txt
echo "== ops" | tee $O/ops.txt
timeout -k 10 500 python3 tools/bench_ops.py 2>/dev/null | grep -i "blur\|upsample2d\|modulate" | cut -c1-300 | tee -a $O/ops.txt
