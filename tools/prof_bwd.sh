#!/bin/bash
# rocprofv3 passes over the backward benchmark (run on the GPU box through gpurun).  One counter group per pass, each
# pass under its own timeout: a counter set the hardware cannot collect aborts rocprofv3 and then hangs in finalisation.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BWD_TORCH=0
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bwd_stats -o run -- python3 $R/tools/bench_bwd.py 4 128 > $R/gpurun_out/bwd_stats.log 2>&1
timeout 120 rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_BUSY_sum TCC_REQ_sum -d $R/gpurun_out/bwd_pmc_a -o run -- python3 $R/tools/bench_bwd.py 4 128 > $R/gpurun_out/bwd_pmc_a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $R/gpurun_out/bwd_pmc_b -o run -- python3 $R/tools/bench_bwd.py 4 128 > $R/gpurun_out/bwd_pmc_b.log 2>&1
