#!/bin/bash
# Builds and runs tools/probes/valu_issue_probe.hip on the GPU box (through gpurun) -> gpurun_out/valu_issue_probe.json + .txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $R/tools/probes/valu_issue_probe.hip -o /tmp/valu_issue_probe || exit 1
timeout -k 10 300 /tmp/valu_issue_probe $R/gpurun_out/valu_issue_probe.json > $R/gpurun_out/valu_issue_probe.txt 2>&1 || { tail -5 $R/gpurun_out/valu_issue_probe.txt; exit 1; }
cat $R/gpurun_out/valu_issue_probe.txt
