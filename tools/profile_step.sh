#!/bin/bash
# Kernel statistics of a short bench run (gpurun -- 'bash tools/profile_step.sh'); prints the per-kernel averages.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_step
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_step -- python3 $R/bench.py --steps 400 --warmup 40 --no-cpu-baseline --prewarm-seconds 0.1 > $R/gpurun_out/prof_step.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/prof_step/runc/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage'])>1: print(r['Name'][:45], r['Calls'], r['AverageNs'])
PY
grep -h '"value"' $R/gpurun_out/prof_step.log | cut -c1-130
