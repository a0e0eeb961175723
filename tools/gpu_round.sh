#!/bin/bash
# One GPU-box session: gpu test-suite (single process, as the driver runs it), smoke, bench, rocprofv3.
# usage: tools/gpu_round.sh [quick|full]   (quick: no PMC passes, no eager bench)
MODE=${1:-full}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
echo "==== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -x --timeout=600 2>&1 | tail -15
echo "==== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "==== bench (graph)"; timeout 600 python bench.py --steps 100 --warmup 10 2> gpurun_out/bench_graph.err | tee gpurun_out/bench_graph.json | cut -c1-700; tail -4 gpurun_out/bench_graph.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/bench_graph.json'))
    print('value', d['value'], 'ms', d['ms_per_step'], 'cpu', d.get('cpu_baseline', {}).get('value'))
    for k in d['kernels']:
        print('  %-32s x%-5.1f avg %8.2f us  %7.3f ms/step  %6.2f TF/s  %7.1f GB/s' % (k['kernel'], k['launches_per_step'], k['avg_us'], k['ms_per_step'], k['tflops'], k['algorithmic_gbs']))
except Exception as e:
    print('no bench json', e)
PY
if [ "$MODE" = "full" ]; then
echo "==== bench (eager)"; timeout 300 python bench.py --steps 50 --warmup 5 --no-graph --no-cpu-baseline 2> gpurun_out/bench_eager.err | tee gpurun_out/bench_eager.json | cut -c1-300
fi
echo "==== rocprofv3 kernel-trace (eager, 20 steps)"
rm -rf gpurun_out/prof/trace
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/trace_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/trace/bench_results.db > gpurun_out/prof/trace_summary.txt; head -32 gpurun_out/prof/trace_summary.txt
if [ "$MODE" = "full" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
echo "==== rocprofv3 pmc $c"
rm -rf gpurun_out/prof/pmc_$c
cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/prof/pmc_${c}_stdout.txt 2>&1; cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/prof/pmc_$c/bench_results.db --json gpurun_out/prof/pmc_$c.json | sed -n '/PMC/,$p' > gpurun_out/prof/pmc_${c}_summary.txt; head -12 gpurun_out/prof/pmc_${c}_summary.txt
done
fi
find gpurun_out/prof -name "*.db" -delete
