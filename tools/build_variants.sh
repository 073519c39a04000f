#!/bin/bash
# Timing-only ablation builds of libgnerf_hip.so (outputs are WRONG by construction; never shipped, never tested
# for parity).  Used with tools/ablate.py to see which part of the render kernel the time goes to.
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/g-nerf_amd/csrc"
out="$root/g-nerf_amd/gnerf_hip/variants"
mkdir -p "$out"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I$src -Wno-unused-value"
for v in "$@"; do
  defs=""
  IFS='+' read -ra parts <<< "$v"
  for p in "${parts[@]}"; do if [ "$p" = STAMPS ]; then defs="$defs -DGNERF_STAMPS"; elif [ "${p#D:}" != "$p" ]; then defs="$defs -D${p#D:}"; elif [ "$p" != base ]; then defs="$defs -DGNERF_ABLATE_$p"; fi; done
  ( /opt/rocm/bin/hipcc $FLAGS $defs -shared "$src"/capi.hip "$src"/planes.hip "$src"/modconv.hip "$src"/render.hip "$src"/bias_act.hip "$src"/upfirdn2d.hip "$src"/filtered_lrelu.hip "$src"/filtered_lrelu_fused.hip "$src"/grid_sample.hip -o "$out/libgnerf_$v.so" && echo "[variant] $v" ) &
done
wait
