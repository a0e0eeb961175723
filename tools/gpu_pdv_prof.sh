#!/bin/bash
# PDV two-stage detector at 8 frames per pass: per-method device time and a kernel trace.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pdv; mkdir -p $O
B=${1:-8}
timeout 300 python tools/bench_pdv.py --math f16x2 --batch $B --reps 5 --phases > $O/phases_b$B.json 2> $O/phases_b$B.txt; cat $O/phases_b$B.txt | tail -20; cat $O/phases_b$B.json
rm -rf $O/trace; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python $GRAFT_REPO_ROOT/tools/bench_pdv.py --math f16x2 --batch $B --reps 5 > $O/trace_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/trace/t_results.db > $O/kernel_trace_pdv_b$B.txt; head -45 $O/kernel_trace_pdv_b$B.txt
find $O -name "*.db" -delete
