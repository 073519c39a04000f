#!/bin/bash
# Build libgnerf_hip.so for gfx950 in-tree (next to the sources' parent: g-nerf_amd/gnerf_hip/).
# hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box alike.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$here/../gnerf_hip/libgnerf_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$root/include -I$here -Wall -Wno-unused-function -Wno-unused-command-line-argument ${GNERF_EXTRA_FLAGS:-}"
. "$here/compile_unit.sh"
objs=()
pids=()
for src in capi bias_act upfirdn2d filtered_lrelu filtered_lrelu_fused grid_sample planes modconv conv3x3 render; do
    [ -f "$here/$src.hip" ] || continue
    obj="$here/$src.o"
    stale=0
    [ -f "$obj" ] || stale=1
    for dep in "$here/$src.hip" "$here"/*.h "$here"/*.inl "$root/include/gnerf_hip.h" "$here/build.sh" "$here/compile_unit.sh" "$here/pk_opsel_fixup.py"; do
        [ "$stale" = 1 ] || { [ "$dep" -nt "$obj" ] && stale=1; } || true
    done
    if [ "$stale" = 1 ]; then
        echo "[build] $src.hip"
        rm -f "$obj"                     # a failed compile must not leave a stale object for the link
        compile_unit "$src" "$obj" &
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
# which commit this library was built from (profiles and bench lines quote it; the GPU box has no .git)
if git -C "$root" rev-parse --short HEAD > /dev/null 2>&1; then
    echo "$(git -C "$root" rev-parse --short HEAD)$(git -C "$root" diff --quiet HEAD -- g-nerf_amd include 2> /dev/null || echo '+local changes')" > "$here/../gnerf_hip/BUILD_HEAD"
fi

# ---- the thin PyTorch-ROCm C++ extension over the same C ABI (host code only: g++, no device code)
ext="$here/../gnerf_hip/gnerf_torch_ext.so"
PY="${PYTHON:-python3}"
stale=0
[ -f "$ext" ] || stale=1
for dep in "$here/torch_binding.cpp" "$root/include/gnerf_hip.h" "$here/build.sh"; do
    [ "$stale" = 1 ] || { [ "$dep" -nt "$ext" ] && stale=1; } || true
done
if [ "$stale" = 1 ]; then
    echo "[build] torch_binding.cpp"
    read -r TORCH_DIR PY_INC CXX11 < <($PY -c "import os, sysconfig, torch; print(os.path.dirname(torch.__file__), sysconfig.get_paths()['include'], int(torch._C._GLIBCXX_USE_CXX11_ABI))")
    rm -f "$ext"
    g++ -O2 -std=c++17 -fPIC -shared -w -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -DTORCH_EXTENSION_NAME=gnerf_torch_ext \
        -DTORCH_API_INCLUDE_EXTENSION_H -D_GLIBCXX_USE_CXX11_ABI=$CXX11 \
        -I"$root/include" -I"$TORCH_DIR/include" -I"$TORCH_DIR/include/torch/csrc/api/include" -I/opt/rocm/include -I"$PY_INC" \
        "$here/torch_binding.cpp" -o "$ext" \
        -L"$TORCH_DIR/lib" -lc10 -lc10_hip -ltorch -ltorch_cpu -ltorch_hip -ltorch_python \
        -L"$here/../gnerf_hip" -lgnerf_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$TORCH_DIR/lib"
    echo "[build] $ext"
fi
