#!/bin/bash
# Round-6 experiment 24: the last block's ToRGB in its convolution's epilogue, no layer output (GNERF_FUSED_TORGB, default 1) against the three launches (0):
# the orbit frame by frame (tools/bench_generator.py) and bench.py's secondary line (8 views per call), alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp24
mkdir -p $O
for v in 1 0 1 0; do
  export GNERF_FUSED_TORGB=$v
  timeout -k 10 300 python3 tools/bench_generator.py 2>/dev/null | grep '"config": 4' | head -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print(json.dumps({'fused_torgb': $v, 'frame_by_frame_eager': d['frames_per_s'], 'frame_by_frame_graph': d['graph_frames_per_s']}))" | tee -a $O/fused_torgb_ab.jsonl || exit 1
  timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 --reps 3 --no-cpu-baseline --no-backward 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['secondary']
print(json.dumps({'fused_torgb': $v, 'views8_graph': round(d['hip_graph_views_value'], 1), 'views8_eager': round(d['eager_views_value'], 1), 'graph': round(d['hip_graph_value'], 1), 'eager': round(d['eager_value'], 1)}))" | tee -a $O/fused_torgb_ab.jsonl || exit 1
done
