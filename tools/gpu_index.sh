#!/bin/bash
# voxelize + index chain alone: wall time and the rocprofv3 kernel trace.  usage: tools/gpu_index.sh [tag]
TAG=${1:-idx}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
timeout 300 python tools/bench_index.py 2>/dev/null | tail -1 | tee $O/${TAG}_bench_index.json
rm -rf $O/trace_index; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_index -o t -- python $GRAFT_REPO_ROOT/tools/bench_index.py --reps 10 > $O/trace_index_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/trace_index/t_results.db > $O/${TAG}_kernel_trace_index.txt; head -40 $O/${TAG}_kernel_trace_index.txt
python tools/trace_timeline.py $O/trace_index/t_results.db 14 k_level_keys > $O/${TAG}_timeline_index.txt; tail -3 $O/${TAG}_timeline_index.txt
find $O -name "*.db" -delete
