#!/bin/bash
# Round-3 evidence set on one GPU box: gpu test-suite, smoke, the default bench line (all legs), rocprofv3 kernel traces of the
# headline, fp32, multisweep and refiner workloads (each its own command), HBM-side PMC passes of the headline command (separate
# passes), SQ matrix-pipe counters.
# usage: tools/gpu_round3.sh [tag] [skip-tests]     outputs -> gpurun_out/r03/<tag>_*
TAG=${1:-r03a}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
if [ -z "$2" ]; then
echo "==== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -4 | tee $O/${TAG}_gputests.txt
echo "==== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/${TAG}_smoke.txt
fi
echo "==== bench (default command)"; timeout 900 python bench.py 2> $O/${TAG}_bench.err > $O/${TAG}_bench_graph.json; tail -12 $O/${TAG}_bench.err
COMMON="--steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux --no-refine --no-pdv"
trace() {   # name, args...
  local name=$1; shift
  echo "==== rocprofv3 kernel-trace: $name ($*)"
  rm -rf $O/trace_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $COMMON "$@" > $O/trace_${name}_stdout.txt 2>&1 )
  python tools/rocpd_summary.py $O/trace_$name/t_results.db > $O/${TAG}_kernel_trace_$name.txt; head -14 $O/${TAG}_kernel_trace_$name.txt
}
trace bench_eager20
trace fp32 --math f32 --batch 8
trace multisweep --sweeps 2 --batch 8
PARGS="--steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux --no-refine --no-pdv"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "==== rocprofv3 pmc $c"
  rm -rf $O/pmc_$c; ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $GRAFT_REPO_ROOT/bench.py $PARGS > $O/pmc_${c}_stdout.txt 2>&1 )
  python tools/rocpd_summary.py $O/pmc_$c/bench_results.db --json $O/pmc_$c.json | sed -n '/PMC/,$p' > $O/${TAG}_pmc_${c}_bench_eager3.txt; head -8 $O/${TAG}_pmc_${c}_bench_eager3.txt
done
python - <<PY
import json
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    try:
        for k, v in json.load(open('$O/pmc_%s.json' % c)).items():
            out.setdefault(k, {}).update(v)
    except Exception as e:
        print('no', c, e)
json.dump(out, open('$O/${TAG}_pmc_traffic.json', 'w'), indent=1, sort_keys=True)
print('traffic entries', len(out))
PY
echo "==== rocprofv3 pmc SQ (matrix pipe)"
rm -rf $O/pmc_sq; ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace -d $O/pmc_sq -o bench -- python $GRAFT_REPO_ROOT/bench.py $PARGS > $O/pmc_sq_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/pmc_sq/bench_results.db | sed -n '/PMC/,$p' > $O/${TAG}_pmc_SQ_bench_eager3.txt; grep -E "MFMA_BUSY" $O/${TAG}_pmc_SQ_bench_eager3.txt | head -8
echo "==== refiner"
for m in f32 f16x2; do timeout 300 python tools/bench_refine.py --math $m 2>/dev/null | tail -1 > $O/${TAG}_bench_refine_$m.json; cut -c1-400 $O/${TAG}_bench_refine_$m.json; done
rm -rf $O/trace_refine; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_refine -o refine -- python $GRAFT_REPO_ROOT/tools/bench_refine.py --math f16x2 > $O/trace_refine_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/trace_refine/refine_results.db > $O/${TAG}_kernel_trace_refine.txt; head -14 $O/${TAG}_kernel_trace_refine.txt
echo "==== two-stage detector (PDV second stage)"
timeout 300 python tools/bench_pdv.py 2>/dev/null | tail -1 > $O/${TAG}_bench_pdv.json; cat $O/${TAG}_bench_pdv.json
rm -rf $O/trace_pdv; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_pdv -o pdv -- python $GRAFT_REPO_ROOT/tools/bench_pdv.py --reps 5 > $O/trace_pdv_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/trace_pdv/pdv_results.db > $O/${TAG}_kernel_trace_pdv.txt; head -12 $O/${TAG}_kernel_trace_pdv.txt
echo "==== voxelize + index chain on its own"
timeout 300 python tools/bench_index.py 2>/dev/null | tail -1 > $O/${TAG}_bench_index.json; cat $O/${TAG}_bench_index.json
find $O -name "*.db" -delete
