#!/bin/bash
# One GPU-box session: full gpu test-suite (single process, as the driver runs it), smoke, bench, rocprofv3.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
echo "==== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -8
echo "==== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "==== bench (graph)"; timeout 600 python bench.py --steps 100 --warmup 10 2> gpurun_out/bench_graph.err | tee gpurun_out/bench_graph.json | cut -c1-900; tail -8 gpurun_out/bench_graph.err
echo "==== bench (eager)"; timeout 300 python bench.py --steps 50 --warmup 5 --no-graph --no-cpu-baseline 2> gpurun_out/bench_eager.err | tee gpurun_out/bench_eager.json | cut -c1-400
echo "==== rocprofv3 kernel-trace stats (eager, 20 steps)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/trace_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof/trace | head -20
f=$(find gpurun_out/prof/trace -name "*kernel_stats*.csv" | head -1); echo "stats file: $f"; head -40 "$f" | cut -c1-200
echo "==== rocprofv3 pmc FETCH_SIZE"
cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_fetch_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
echo "==== rocprofv3 pmc WRITE_SIZE"
cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_write_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*.csv" | head; du -sh gpurun_out/prof
# keep the merged output small: drop raw traces > 20 MB
find gpurun_out/prof -size +20M -delete
