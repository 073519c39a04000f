#!/bin/bash
# Builds and runs tools/probes/pk_opsel_hazard_probe.hip on the GPU box (through gpurun) -> gpurun_out/pk_opsel_hazard_probe.{json,txt}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/probes/pk_opsel_hazard_probe.hip -o /tmp/pk_opsel_hazard_probe 2> /dev/null || exit 1
timeout -k 10 300 /tmp/pk_opsel_hazard_probe $R/gpurun_out/pk_opsel_hazard_probe.json ${1:-100000} > $R/gpurun_out/pk_opsel_hazard_probe.txt 2>&1 || { tail -5 $R/gpurun_out/pk_opsel_hazard_probe.txt; exit 1; }
cut -c1-230 $R/gpurun_out/pk_opsel_hazard_probe.txt
