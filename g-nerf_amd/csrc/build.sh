#!/bin/bash
# Build libgnerf_hip.so for gfx950 in-tree (next to the sources' parent: g-nerf_amd/gnerf_hip/).
# hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box alike.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$here/../gnerf_hip/libgnerf_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I$here -Wall -Wno-unused-function ${GNERF_EXTRA_FLAGS:-}"
objs=()
pids=()
for src in capi bias_act upfirdn2d filtered_lrelu filtered_lrelu_fused grid_sample planes render; do
    [ -f "$here/$src.hip" ] || continue
    obj="$here/$src.o"
    stale=0
    [ -f "$obj" ] || stale=1
    for dep in "$here/$src.hip" "$here"/*.h "$here"/*.inl "$root/include/gnerf_hip.h" "$here/build.sh"; do
        [ "$stale" = 1 ] || { [ "$dep" -nt "$obj" ] && stale=1; } || true
    done
    if [ "$stale" = 1 ]; then
        echo "[build] $src.hip"
        rm -f "$obj"                     # a failed compile must not leave a stale object for the link
        $HIPCC $FLAGS -c "$here/$src.hip" -o "$obj" &
        pids+=($!)
    fi
    objs+=("$obj")
done
# a bare `wait` returns 0 whatever the children did: wait for each compile and stop at the first failure
for pid in "${pids[@]}"; do
    wait "$pid" || { echo "[build] a compile failed" >&2; exit 1; }
done
$HIPCC -shared -fPIC --offload-arch=gfx950 "${objs[@]}" -o "$out"
echo "[build] $out"
