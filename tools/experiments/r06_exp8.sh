#!/bin/bash
# Round-6 experiment 8: the step's rays and draws in one launch (gnerf_make_rays_and_draws): equality with the three launches, then the headline step
# with it on and off on one box.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp8
mkdir -p $O
echo "build $(cat g-nerf_amd/gnerf_hip/BUILD_HEAD)" | tee $O/tests.txt
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rays_and_draws or philox or make_rays" 2>&1 | tail -6 | tee -a $O/tests.txt
grep -q "failed\|error\|core dump" $O/tests.txt && { echo "stopping" | tee -a $O/tests.txt; exit 1; }
: > $O/step_ab.jsonl
for rep in 1 2; do
for f in 1 0; do
  GNERF_BENCH_FUSED_PREP=$f timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-backward 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'fused_prep': $f, 'value_Mrays': round(d['value'] / 1e6, 2), 'ms_per_step': round(d['ms_per_step'], 4), 'render_call_ms': d['roofline']['render_call_ms']['auto'], 'frac': round(d['roofline']['frac'], 4), 'producer_layout_Mrays': round(d['config'].get('producer_layout_value', 0) / 1e6, 2)}))" | tee -a $O/step_ab.jsonl
done
done
