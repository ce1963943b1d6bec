#!/bin/bash
# Emulated 1/G shard step of a workload with the per-launch breakdown (rocprofv3 kernel statistics), one shard count and front per
# profiler run:   gpurun --timeout 1200 -- 'bash tools/dp_shards.sh c4 "1 8" "replicated sharded"'
# Writes gpurun_out/shards/<workload>_G<g>_<front>.{log,csv}; copy the csv files worth keeping into profiles/.
R=$GRAFT_REPO_ROOT
WL=${1:-c4}; GS=${2:-"1 2 4 8"}; FRONTS=${3:-"replicated sharded"}; PREC=${4:-}
cd /tmp && export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
mkdir -p $R/gpurun_out/shards
for G in $GS; do for F in $FRONTS; do
  [ "$G" = 1 ] && [ "$F" = sharded ] && continue
  tag=${WL}${PREC:+_$PREC}_G${G}_$F
  rm -rf $R/gpurun_out/shards/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/shards/$tag -- python3 $R/tools/dp_selftest.py --workload $WL --shards $G \
      --fronts $F --schedules plain --skip-check ${PREC:+--precision $PREC} > $R/gpurun_out/shards/$tag.log 2>&1 || { tail -5 $R/gpurun_out/shards/$tag.log; exit 1; }
  grep -h "emulated shard" $R/gpurun_out/shards/$tag.log
  cp $(ls $R/gpurun_out/shards/$tag/*/*kernel_stats.csv | head -1) $R/gpurun_out/shards/$tag.csv
  python3 - $R/gpurun_out/shards/$tag.csv <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "vgan::" in r["Name"] or "nccl" in r["Name"].lower() or "rccl" in r["Name"].lower()]
calls = max(int(r["Calls"]) for r in rows)
for r in rows:
    if int(r["Calls"]) * 4 >= calls:
        print("   %-64s %6s calls %9.1f us" % (r["Name"].replace("void ", "")[:64], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done; done
