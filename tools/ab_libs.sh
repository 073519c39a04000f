#!/bin/bash
# A/B of variant libraries on one box: tools/ab_libs.sh "<command>" default <variant> [...]   (variants from tools/build_variants.sh)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cmd=$1; shift
for v in "$@"; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB=$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so; fi
  echo "== $v"
  eval "$cmd" 2>/dev/null | cut -c1-400
done
