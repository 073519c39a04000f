#!/bin/bash
# Variant builds of libgnerf_hip.so for A/B timing and ablation (g-nerf_amd/gnerf_hip/variants/libgnerf_<v>.so, selected with
# GNERF_HIP_LIB): render.hip (or $VARIANT_UNIT.hip) recompiled with the variant's macros through the same assembly pass as the product build, the other
# objects taken from the product build (run g-nerf_amd/csrc/build.sh first).  A variant is a '+'-joined list of parts: `base`,
# `STAMPS` (-DGNERF_STAMPS), `D:MACRO[=v]` (-DMACRO[=v]), anything else X -> -DGNERF_ABLATE_X (timing only: outputs are WRONG).
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
here="$root/g-nerf_amd/csrc"
out="$root/g-nerf_amd/gnerf_hip/variants"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I$here -Wno-unused-value -Wno-unused-command-line-argument"
. "$here/compile_unit.sh"
unit="${VARIANT_UNIT:-render}"          # the translation unit that is recompiled with the variant's macros
for v in "$@"; do
  defs=""
  IFS='+' read -ra parts <<< "$v"
  for p in "${parts[@]}"; do if [ "$p" = STAMPS ]; then defs="$defs -DGNERF_STAMPS"; elif [ "${p#D:}" != "$p" ]; then defs="$defs -D${p#D:}"; elif [ "$p" != base ]; then defs="$defs -DGNERF_ABLATE_$p"; fi; done
  ( compile_unit $unit "$out/${unit}_$v.o" "$defs" > /dev/null
    others=(); for u in capi bias_act upfirdn2d filtered_lrelu filtered_lrelu_fused grid_sample planes modconv conv3x3 render; do [ $u = $unit ] || others+=("$here/$u.o"); done
    $HIPCC -shared -fPIC --offload-arch=gfx950 "${others[@]}" "$out/${unit}_$v.o" -o "$out/libgnerf_$v.so"
    python3 "$root/tools/kernel_resources.py" "$out/${unit}_$v.o" > "$out/resources_$v.txt" 2>/dev/null || true
    rm -f "$out/${unit}_$v.o"; echo "[variant] $v" ) &
done
wait
