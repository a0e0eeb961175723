#!/bin/bash
# development: conv3x3 diag variants under SQ counters (clock = SQ_BUSY_CU_CYCLES / 256 / duration) + MFMA micro-benchmark
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/c3diag
mkdir -p $O
./gpurun_bin_mfma_peak 2>&1 | tee $O/mfma_peak.txt
for d in ${@:-0 4 8 15 16}; do
  rm -rf $O/p$d
  ( cd /tmp && DZ_TUNE_C3_DIAG=$d timeout 200 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $O/p$d -o c3 -- python $GRAFT_REPO_ROOT/tools/bench_conv3x3.py --iters 10 > $O/p${d}_stdout.txt 2>&1 )
  tail -1 $O/p${d}_stdout.txt
  python tools/rocpd_summary.py $O/p$d/c3_results.db | grep -E "k_conv3x3" | head -12
done
find $O -name "*.db" -delete
