#!/bin/bash
# One round of evidence on a GPU box (run through gpurun from the repo root), in two calls that each fit gpurun's limit:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03 a'     tests, bench lines of every BASELINE.json configuration,
#                                                                     rocprofv3 kernel statistics + PMC passes of the c3 step
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r03 b'     PMC passes at c4 / c5, emulated 1/G shard steps (c3, c4, c5)
#                                                                     with the per-launch breakdown of G = 1 and G = 8
# Everything lands under gpurun_out/<tag>_*; copy what is to be kept into profiles/.  (rocprofv3 gets the program itself after
# `--`, never a wrapper, and --pmc passes carry no trace domain: see the pool rules.)
set -o pipefail
TAG=${1:-rXX}
PART=${2:-a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
if [ "$PART" = a ]; then
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; tail -3 $O/${TAG}_pytest.log; [ $rc -eq 0 ] || exit 1
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
cut -c1-300 $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_prof $O/${TAG}_pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extra > $O/${TAG}_prof.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --output-format csv -d $O/${TAG}_pmc_$c -- python3 $R/tools/kbench.py > $O/${TAG}_pmc_$c.log 2>&1 || exit 1
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/${TAG}_pmc_SQ -- python3 $R/tools/kbench.py > $O/${TAG}_pmc_SQ.log 2>&1 || exit 1
cd $R
python3 tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $O/${TAG}_pmc_SQ > /dev/null && cp profiles/traffic.json $O/${TAG}_traffic.json
cp $(ls $O/${TAG}_prof/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats.csv
for w in c1 c2 c4 c5; do python bench.py --workload $w > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err; cut -c1-160 $O/${TAG}_bench_$w.json; done
python bench.py --workload c2 --precision bf16x3 > $O/${TAG}_bench_c2_bf16x3.json 2> $O/${TAG}_bench_c2_bf16x3.err; cut -c1-160 $O/${TAG}_bench_c2_bf16x3.json
python bench.py --workload c5 --precision fp32 > $O/${TAG}_bench_c5_fp32.json 2> $O/${TAG}_bench_c5_fp32.err; cut -c1-160 $O/${TAG}_bench_c5_fp32.json
python bench.py --path kl > $O/${TAG}_bench_kl.json 2> $O/${TAG}_bench_kl.err; cut -c1-160 $O/${TAG}_bench_kl.json
else
bash tools/pmc_c5.sh c5 | tail -1
bash tools/pmc_c5.sh c4 | tail -1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
( timeout -k 10 300 python tools/dp_selftest.py --workload c3 --shards 1,2,4,8 2>&1 | grep -E "emulated|graph="
  for w in c4 c5; do timeout -k 10 300 python tools/dp_selftest.py --workload $w --shards 1,2,4,8 --fronts replicated,sharded --skip-check 2>&1 | grep emulated; done
  timeout -k 10 300 python tools/dp_selftest.py --workload c5 --precision fp32 --shards 1,2,4,8 --fronts replicated,sharded --skip-check 2>&1 | grep emulated ) | tee $O/${TAG}_shards.txt
for w in c4 c5; do bash tools/dp_shards.sh $w "1 8" "replicated sharded" > $O/${TAG}_shards_$w.log 2>&1; done
bash tools/dp_shards.sh c5 "1 8" "replicated sharded" fp32 > $O/${TAG}_shards_c5_fp32.log 2>&1
for t in c4_G8_sharded c5_G8_sharded c5_fp32_G8_sharded c5_G1_replicated; do python3 tools/step_timeline.py $O/shards/$t > $O/${TAG}_timeline_$t.txt; done
fi
echo done
