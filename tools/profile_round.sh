#!/bin/bash
# One round of evidence on a GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# GPU test tier, the default bench line, the rocprofv3 kernel statistics of a bench run, the separate PMC passes that
# tools/pmc_traffic.py folds into profiles/traffic.json, and the bench lines of the other BASELINE.json configurations.
# Everything lands under gpurun_out/<tag>_*; copy what is to be kept into profiles/.  (rocprofv3 gets the program itself after
# `--`, never a wrapper: see the pool rules.)
set -o pipefail
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; tail -3 $O/${TAG}_pytest.log; [ $rc -eq 0 ] || exit 1
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
cut -c1-400 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_prof $O/${TAG}_pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $O/${TAG}_prof.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -- python3 $R/tools/kbench.py > $O/${TAG}_pmc_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_pmc_SQ -- python3 $R/tools/kbench.py > $O/${TAG}_pmc_SQ.log 2>&1 || exit 1
cd $R
for w in c1 c2 c4; do python bench.py --workload $w > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err; cut -c1-200 $O/${TAG}_bench_$w.json; done
python bench.py --workload c5 > $O/${TAG}_bench_c5.json 2> $O/${TAG}_bench_c5.err; cut -c1-200 $O/${TAG}_bench_c5.json
python bench.py --workload c5 --precision fp32 > $O/${TAG}_bench_c5_fp32.json 2> $O/${TAG}_bench_c5_fp32.err; cut -c1-200 $O/${TAG}_bench_c5_fp32.json
python bench.py --path kl > $O/${TAG}_bench_kl.json 2> $O/${TAG}_bench_kl.err; cut -c1-200 $O/${TAG}_bench_kl.json
echo done
