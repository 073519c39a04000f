#!/bin/bash
# Round-6 experiment 16: where a convolution workgroup's life goes (shipped eight-wave form): in-kernel stamps before the first loads, behind the
# prologue's barrier, at the main loop's end, in front of the store loop and at the end.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp16
mkdir -p $O
GNERF_HIP_LIB=$R/g-nerf_amd/gnerf_hip/variants/libgnerf_D:GNERF_CONV_STAMPS.so timeout -k 10 200 python3 tools/conv_clock.py 2>/dev/null | grep '^{' | tee $O/conv_segments.jsonl
