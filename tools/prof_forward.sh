#!/bin/bash
# rocprofv3 passes over bench.py (run on the GPU box through gpurun):   bash tools/prof_forward.sh <tag>
# One counter group per pass and every pass under its own timeout (a counter set the hardware cannot collect aborts
# rocprofv3, which then hangs in finalisation).  Results (rocpd sqlite) land in gpurun_out/prof_<tag>/<pass>/;
# tools/prof_collect.py turns them into the files kept under profiles/.
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_$tag
mkdir -p $out
CMD="python3 $R/bench.py --steps 20 --warmup 3 --reps 3 --no-cpu-baseline --no-secondary"
timeout 180 rocprofv3 --kernel-trace --stats -d $out/stats -o run -- $CMD > $out/stats.log 2>&1
pass() { name=$1; shift; timeout 180 rocprofv3 --pmc "$@" -d $out/$name -o run -- $CMD > $out/$name.log 2>&1; }
pass sq1 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
pass sq2 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
cd $R && python3 tools/prof_collect.py $out $tag && mkdir -p gpurun_out/profiles && cp profiles/${tag}_* profiles/traffic.json gpurun_out/profiles/
