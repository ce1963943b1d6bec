R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in abf3_base abf3_nolds ablate_base; do
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc17_$v -- $R/tools/bin/$v 512 > $R/gpurun_out/pmc17_$v.log 2>&1 || { tail -5 $R/gpurun_out/pmc17_$v.log; exit 1; }
done
echo ok
