#!/bin/bash
# PMC pass over the eager bench (3 steps): SQ counters per kernel.  usage: tools/gpu_pmc.sh "<counters>" tag
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
C="$1"; TAG=${2:-sq}
mkdir -p gpurun_out/prof
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+|TCC_[A-Z0-9_]+|TCP_[A-Z0-9_]+|GRBM_[A-Z_]+" | sort -u > gpurun_out/prof/counters_avail.txt
wc -l gpurun_out/prof/counters_avail.txt
rm -rf gpurun_out/prof/pmc_$TAG
cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_${TAG}_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/prof/pmc_${TAG}_stdout.txt
python tools/rocpd_summary.py gpurun_out/prof/pmc_$TAG/bench_results.db | sed -n '/PMC/,$p' > gpurun_out/prof/pmc_${TAG}_summary.txt
grep -E "k_spconv|k_conv2d" gpurun_out/prof/pmc_${TAG}_summary.txt | head -80
find gpurun_out/prof -name "*.db" -delete
