#!/bin/bash
# profile set of a round without the test suite: bench line, kernel trace, PMC FETCH/WRITE/SQ passes, graph-step timeline
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 600 python bench.py 2> gpurun_out/bench_graph.err > gpurun_out/bench_graph.json; tail -2 gpurun_out/bench_graph.err
rm -rf gpurun_out/prof/trace
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/trace_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/bench_results.db > gpurun_out/prof/trace_summary.txt; head -12 gpurun_out/prof/trace_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf gpurun_out/prof/pmc_$c
cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_${c}_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/pmc_$c/bench_results.db --json gpurun_out/prof/pmc_$c.json | sed -n '/PMC/,$p' > gpurun_out/prof/pmc_${c}_summary.txt; head -6 gpurun_out/prof/pmc_${c}_summary.txt
done
find gpurun_out/prof -name "*.db" -delete
bash tools/gpu_pmc.sh "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" sqh > gpurun_out/prof/pmc_sqh.log 2>&1
bash tools/gpu_trace_graph.sh | tail -1
