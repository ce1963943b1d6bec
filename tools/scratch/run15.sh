timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bf3 or bf16x3" 2>&1 | tail -2 || exit 1
cd /tmp && export TMPDIR=/tmp && VGAN_BF3_BK=64 VGAN_MMD_PRECISION=bf16x3 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof15 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline --prewarm-seconds 0.1 > $GRAFT_REPO_ROOT/gpurun_out/prof15.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof15/runc/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage'])>1: print(r['Name'][:45], r['Calls'], r['AverageNs'])
PY
grep -h '"value"' $GRAFT_REPO_ROOT/gpurun_out/prof15.log | cut -c1-130
