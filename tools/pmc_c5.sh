#!/bin/bash
# PMC passes of the MMD kernels at c5 (d=4096, batch=8192): gpurun --timeout 1200 -- 'bash tools/pmc_c5.sh'
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && export VGAN_KBENCH_WORKLOAD=c5
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc5_$c -- python3 $R/tools/kbench.py > $R/gpurun_out/pmc5_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc5_SQ -- python3 $R/tools/kbench.py > $R/gpurun_out/pmc5_SQ.log 2>&1 || exit 1
echo done
