#!/bin/bash
# Round-6 experiment 4: config 3 kernel by kernel (marker-bracketed, solver search excluded), fp32-grade convolutions on and off.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp4
mkdir -p $O
for f in 1 0; do
  export GNERF_F32X3=$f
  MARKED_SCRIPT=tools/config3_marked.py bash tools/prof_orbit.sh r06_config3_f32x3_$f > $O/config3_f32x3_$f.txt 2>&1
  cp gpurun_out/r06_config3_f32x3_${f}_kernel_stats.csv gpurun_out/r06_config3_f32x3_${f}_summary.json $O/ 2>/dev/null
  head -c 2500 $O/config3_f32x3_$f.txt
done
unset GNERF_F32X3
head -30 $O/r06_config3_f32x3_1_kernel_stats.csv | cut -c1-200
