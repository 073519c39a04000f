#!/bin/bash
# Round-6 experiment 23: the plain convolution without its output (timing-only CONVNOSTORE: the epilogue's values are computed and dropped -- no LDS
# staging, no store loop) against the shipped kernel: the bound on what fusing the last block's ToRGB into the epilogue could save on the convolution's side.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
O=$R/gpurun_out/r06_exp23
mkdir -p $O
V=$R/g-nerf_amd/gnerf_hip/variants
for v in base CONVNOSTORE base CONVNOSTORE; do
  echo "== $v" | tee -a $O/conv_nostore.txt
  GNERF_HIP_LIB=$V/libgnerf_$v.so timeout -k 10 300 python3 - <<'PY' 2>/dev/null | tee -a $O/conv_nostore.txt
import os, sys, json
sys.path[:0] = ['g-nerf_amd', '.']
import torch, gnerf_hip
dev = torch.device('cuda', 0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)
g = torch.Generator(device='cpu').manual_seed(1)
for (n, cin, cout, h, w) in [(4, 128, 128, 512, 512), (8, 128, 128, 512, 512)]:
    x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
    wpk = gnerf_hip.pack_conv3x3_weights((torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev))
    dco = (torch.rand(n, cout, generator=g) + 0.5).to(dev); bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    ms = timeit(lambda: gnerf_hip.conv3x3_epilogue(x, wpk, bias, scale=dco, gain=2 ** 0.5, clamp=256.0))
    print(json.dumps({'shape': [n, cin, cout, h, w], 'ms': round(ms, 4)}))
PY
done
