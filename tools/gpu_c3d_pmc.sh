cd $GRAFT_REPO_ROOT
for v in 0 1; do DZ_TUNE_C3_D=$v timeout 120 python tools/bench_conv3x3.py --data relu --batch 32 2>&1 | tail -1; done
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c3d; mkdir -p $O
for v in 0 1; do
rm -rf $O/p$v
( cd /tmp && DZ_TUNE_C3_D=$v timeout 200 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $O/p$v -o c3 -- python $GRAFT_REPO_ROOT/tools/bench_conv3x3.py --iters 10 --data relu --batch 32 > $O/p${v}_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/p$v/c3_results.db | grep -E "k_conv3x3" | head -8
rm -rf $O/q$v
( cd /tmp && DZ_TUNE_C3_D=$v timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/q$v -o c3 -- python $GRAFT_REPO_ROOT/tools/bench_conv3x3.py --iters 10 --data relu --batch 32 > $O/q${v}_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/q$v/c3_results.db | grep -E "k_conv3x3" | grep -v "  calls  " | tail -4
done
find $O -name "*.db" -delete
