#!/bin/bash
# Runs tools/dbg_bwd_stage.py once per variant build of the library (g-nerf_amd/gnerf_hip/variants/libgnerf_<v>.so), one process each.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
flags=${FLAGS:-8}
for v in "$@"; do
  echo "== $v"
  GNERF_HIP_LIB=$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so timeout -k 10 120 python tools/dbg_bwd_stage.py $flags 2>&1 | grep '"flags"' || echo "FAILED $v"
done
