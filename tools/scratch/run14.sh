timeout -k 10 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "bf3 or bf16x3" 2>&1 | tail -2 || exit 1
for sp in 1 2 3 4; do
cd /tmp && export TMPDIR=/tmp && VGAN_BF3_BK=64 VGAN_BWD_SPLITS=$sp VGAN_MMD_PRECISION=bf16x3 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof14_$sp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline --prewarm-seconds 0.1 > $GRAFT_REPO_ROOT/gpurun_out/prof14_$sp.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof14_$sp/runc/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if 'backward_bf3' in r['Name'] or 'mask_backward' in r['Name']: print($sp, r['Name'][:45], r['Calls'], r['AverageNs'])
PY
done
