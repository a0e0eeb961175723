#!/bin/bash
# HBM-side counters of the voxelize + index chain on its own (separate passes): FETCH_SIZE, WRITE_SIZE per kernel.
# usage: tools/gpu_pmc_index.sh [tag]
TAG=${1:-idx}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_index_$c; ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_index_$c -o t -- python $GRAFT_REPO_ROOT/tools/bench_index.py --reps 3 > $O/pmc_index_${c}_stdout.txt 2>&1 )
  python tools/rocpd_summary.py $O/pmc_index_$c/t_results.db | sed -n '/PMC/,$p' > $O/${TAG}_pmc_${c}_index.txt; head -16 $O/${TAG}_pmc_${c}_index.txt
done
find $O -name "*.db" -delete
