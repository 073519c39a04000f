#!/bin/bash
# Round-5 experiment 18: where the transposed convolution's time goes -- timing-only builds that run ONE output phase (the other jobs exit at once).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r05_exp18
mkdir -p $O
: > $O/phases.txt
for v in default "D:GNERF_CONVT_ONLY_PHASE=0" "D:GNERF_CONVT_ONLY_PHASE=1" "D:GNERF_CONVT_ONLY_PHASE=2" "D:GNERF_CONVT_ONLY_PHASE=3"; do
  if [ "$v" = default ]; then unset GNERF_HIP_LIB; else export GNERF_HIP_LIB="$R/g-nerf_amd/gnerf_hip/variants/libgnerf_$v.so"; fi
  python3 - "$v" <<'PY' | tee -a $O/phases.txt
import os, sys, json
ROOT = os.getcwd()
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch, gnerf_hip
dev = torch.device('cuda', 0)
x = (torch.randn(4, 256, 256, 256, device=dev) * 0.5).half().contiguous(memory_format=torch.channels_last)
wp = gnerf_hip.pack_conv_transpose3x3_weights(torch.randn(128, 256, 3, 3, device=dev) / 32)
for _ in range(3): gnerf_hip.conv_transpose3x3_s2(x, wp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(3):
    e0.record()
    for _ in range(20): gnerf_hip.conv_transpose3x3_s2(x, wp)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
print(json.dumps({'build': sys.argv[1], 'ms': min(ts)}))
PY
done
