#!/bin/bash
# Round-6 experiment 11: the clock the chip holds under the convolution (four- and eight-wave workgroups, random and all-zero operands) and under the
# render kernel -- in-kernel s_memtime / s_memrealtime stamps of the diagnostic builds.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp11
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
for v in "D:GNERF_CONV_STAMPS" "D:GNERF_CONV_STAMPS+D:GNERF_CONV_WAVES=8"; do
  echo "== $v" | tee -a $O/conv_clock.jsonl
  GNERF_HIP_LIB=$V/libgnerf_$v.so timeout -k 10 200 python3 tools/conv_clock.py 2>/dev/null | grep '^{' | tee -a $O/conv_clock.jsonl || exit 1
  GNERF_HIP_LIB=$V/libgnerf_$v.so timeout -k 10 200 python3 tools/conv_clock.py --zeros 2>/dev/null | grep '^{' | tee -a $O/conv_clock.jsonl || exit 1
done
GNERF_HIP_LIB=$V/libgnerf_STAMPS.so timeout -k 10 200 python3 tools/stamps.py 2>&1 | tail -40 | tee $O/render_stamps.txt
