#!/bin/bash
# kernel trace of the refiner bench (tools/bench_refine.py) -> gpurun_out/prof/refine_trace_summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
M=${1:-f16x2}
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/rtrace
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/rtrace -o rf -- python $GRAFT_REPO_ROOT/tools/bench_refine.py --math $M > $GRAFT_REPO_ROOT/gpurun_out/prof/rtrace_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/rtrace/rf_results.db > gpurun_out/prof/refine_trace_summary_$M.txt
find gpurun_out/prof -name "*.db" -delete
head -40 gpurun_out/prof/refine_trace_summary_$M.txt
