#!/bin/bash
# Round-6 experiment 27: non-temporal output stores of the PLAIN convolution (GNERF_CONV_NT=4; alone the kernel is 1-5 % faster, r06_exp26) in the
# pipeline, where the next kernel reads what was stored: the orbit frame by frame and bench.py's secondary line, libraries alternating on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp27
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
for v in base nt4 base nt4; do
  if [ $v = base ]; then export GNERF_HIP_LIB="$V/libgnerf_base.so"; else export GNERF_HIP_LIB="$V/libgnerf_D:GNERF_CONV_NT=4.so"; fi
  timeout -k 10 300 python3 tools/bench_generator.py 2>/dev/null | grep '"config"' | head -2 | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print(json.dumps({'lib': '$v', 'config': d['config'], 'frames_per_s': d['frames_per_s'], 'graph': d.get('graph_frames_per_s')}))" | tee -a $O/conv_nt_pipeline.jsonl || exit 1
  timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 --reps 3 --no-cpu-baseline --no-backward 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['secondary']
print(json.dumps({'lib': '$v', 'views8_graph': round(d['hip_graph_views_value'], 1), 'views8_eager': round(d['eager_views_value'], 1), 'graph': round(d['hip_graph_value'], 1), 'conv_ms': [round(k['ms'], 4) for k in d['roofline']['kernels']]}))" | tee -a $O/conv_nt_pipeline.jsonl || exit 1
done
