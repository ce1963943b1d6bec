#!/bin/bash
# PMC passes of the MMD kernels at a large workload (default c5: d=4096, batch=8192; c4: d=2048, batch=4096), folded into
# profiles/traffic_<workload>.json:      gpurun --timeout 1200 -- 'bash tools/pmc_c5.sh c5'
# Separate --pmc passes with no trace domain beside them (FETCH_SIZE and WRITE_SIZE do not fit one pass: TCC slots).
R=$GRAFT_REPO_ROOT
WL=${1:-c5}
cd /tmp && export TMPDIR=/tmp && export VGAN_KBENCH_WORKLOAD=$WL
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf $R/gpurun_out/pmc_${WL}_$c
rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${WL}_$c -- python3 $R/tools/kbench.py > $R/gpurun_out/pmc_${WL}_$c.log 2>&1 || exit 1
done
rm -rf $R/gpurun_out/pmc_${WL}_SQ
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_${WL}_SQ -- python3 $R/tools/kbench.py > $R/gpurun_out/pmc_${WL}_SQ.log 2>&1 || exit 1
VGAN_TRAFFIC_JSON=traffic_$WL.json python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_${WL}_FETCH_SIZE $R/gpurun_out/pmc_${WL}_WRITE_SIZE $R/gpurun_out/pmc_${WL}_SQ
cp $R/profiles/traffic_$WL.json $R/gpurun_out/traffic_$WL.json
echo done
