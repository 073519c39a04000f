#!/bin/bash
# mkvariant.sh NAME : dev_NAME.s -> /root/repo/g-nerf_amd/gnerf_hip/variants/libgnerf_NAME.so
set -euo pipefail
cd /tmp/asmx
n=$1
D=render-hip-amdgcn-amd-amdhsa-gfx950
c4=$(sed -n 4p cmds.txt | sed "s/\"$D.o\"/\"dev_$n.o\"/; s/\"$D.s\"/\"dev_$n.s\"/")
c5=$(sed -n 5p cmds.txt | sed "s/\"$D.out\"/\"dev_$n.out\"/; s/\"$D.o\"/\"dev_$n.o\"/; s/\"-save-temps\" //")
c6=$(sed -n 6p cmds.txt | sed "s/-input=$D.out/-input=dev_$n.out/; s/-output=render.hip-hip-amdgcn-amd-amdhsa.hipfb/-output=$n.hipfb/")
c8=$(sed -n 8p cmds.txt | sed "s/\"render.hip-hip-amdgcn-amd-amdhsa.hipfb\"/\"$n.hipfb\"/; s/\"render-host-x86_64-unknown-linux-gnu.bc\"/\"host_$n.bc\"/")
c9=$(sed -n 9p cmds.txt | sed "s/\"render-host-x86_64-unknown-linux-gnu.s\"/\"host_$n.s\"/; s/\"render-host-x86_64-unknown-linux-gnu.bc\"/\"host_$n.bc\"/")
c10=$(sed -n 10p cmds.txt | sed "s/\"render.o\"/\"render_$n.o\"/; s/\"render-host-x86_64-unknown-linux-gnu.s\"/\"host_$n.s\"/")
eval "$c4"; eval "$c5"; eval "$c6"; eval "$c8"; eval "$c9"; eval "$c10"
src=/root/repo/g-nerf_amd/csrc
out=/root/repo/g-nerf_amd/gnerf_hip/variants
mkdir -p $out
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $src/capi.o $src/bias_act.o $src/upfirdn2d.o $src/filtered_lrelu.o $src/filtered_lrelu_fused.o $src/grid_sample.o $src/planes.o $src/modconv.o render_$n.o -o $out/libgnerf_$n.so
rm -f dev_$n.o dev_$n.out $n.hipfb host_$n.bc host_$n.s
echo "[variant] $n"
