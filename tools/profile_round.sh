#!/bin/bash
# One round of evidence on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'
# GPU test tier, the default bench line, the rocprofv3 kernel statistics of a bench run, and the separate PMC passes that
# tools/pmc_traffic.py folds into profiles/traffic.json.  Everything lands under gpurun_out/; copy what is to be kept into
# profiles/.  (rocprofv3 gets the program itself after `--`, never a wrapper: see the pool rules.)
set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest16.log 2>&1; rc=$?; tail -3 $R/gpurun_out/pytest16.log; [ $rc -eq 0 ] || exit 1
python bench.py > $R/gpurun_out/bench16.json 2> $R/gpurun_out/bench16.err || exit 1
cut -c1-600 $R/gpurun_out/bench16.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof16 -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $R/gpurun_out/prof16.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc16_$c -- python3 $R/tools/kbench.py > $R/gpurun_out/pmc16_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc16_SQ -- python3 $R/tools/kbench.py > $R/gpurun_out/pmc16_SQ.log 2>&1 || exit 1
echo done
