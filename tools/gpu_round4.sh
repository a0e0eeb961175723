#!/bin/bash
# Round-4 evidence set on one GPU box: gpu test-suite, smoke, the default bench line (all legs), rocprofv3 kernel trace of the
# headline command, HBM-side PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ matrix-pipe counters of the same command,
# the per-layer sparse benchmark of both engines, the x-run kernel's in-kernel cycle accounting (diag build).
# usage: tools/gpu_round4.sh [tag] [skip-tests] [skip-aux]     outputs -> gpurun_out/r04/<tag>_*
TAG=${1:-r04a}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04
mkdir -p $O
if [ -z "$2" ] || [ "$2" = "-" ]; then
echo "==== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -4 | tee $O/${TAG}_gputests.txt
echo "==== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/${TAG}_smoke.txt
fi
echo "==== bench (default command)"; timeout 900 python bench.py 2> $O/${TAG}_bench.err > $O/${TAG}_bench_graph.json; tail -14 $O/${TAG}_bench.err
COMMON="--steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux --no-refine --no-pdv"
trace() {   # name, args...
  local name=$1; shift
  echo "==== rocprofv3 kernel-trace: $name ($*)"
  rm -rf $O/trace_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$name -o t -- python $GRAFT_REPO_ROOT/bench.py $COMMON "$@" > $O/trace_${name}_stdout.txt 2>&1 )
  python tools/rocpd_summary.py $O/trace_$name/t_results.db > $O/${TAG}_kernel_trace_$name.txt; head -16 $O/${TAG}_kernel_trace_$name.txt
}
trace bench_eager20
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
for k, v in sorted(out.items()):
    if 'spconv' in k or 'conv3x3' in k:
        rd = 2.0 * 1024.0 * v.get('FETCH_SIZE', {'per_call': 0})['per_call']; wr = 1024.0 * v.get('WRITE_SIZE', {'per_call': 0})['per_call']
        print('%-60s read %7.1f MB  write %7.1f MB per launch' % (k[:60], rd / 1e6, wr / 1e6))
PY
echo "==== rocprofv3 pmc SQ (matrix pipe)"
rm -rf $O/pmc_sq; ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_sq -o bench -- python $GRAFT_REPO_ROOT/bench.py $PARGS > $O/pmc_sq_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/pmc_sq/bench_results.db | sed -n '/PMC/,$p' > $O/${TAG}_pmc_SQ_bench_eager3.txt; grep -E "MFMA_BUSY" $O/${TAG}_pmc_SQ_bench_eager3.txt | head -10
echo "==== per-layer sparse benchmark, both engines"
for e in gather xrun; do DZ_TUNE_SPCONV_ENGINE=$e timeout 300 python tools/bench_spconv.py --batch 16 --reps 20 --math f16x2 2>&1 | tail -23 > $O/${TAG}_spconv_layers_$e.txt; tail -1 $O/${TAG}_spconv_layers_$e.txt; done
echo "==== x-run kernel: cycle accounting (diag build)"
timeout 300 bash tools/gpu_x_diag.sh "512" 2>&1 | grep -v "steps/wave *per wave" | tee $O/${TAG}_xrun_cycles.txt
if [ -z "$3" ]; then
echo "==== refiner"
for m in f32 f16x2; do timeout 300 python tools/bench_refine.py --math $m 2>/dev/null | tail -1 > $O/${TAG}_bench_refine_$m.json; cut -c1-500 $O/${TAG}_bench_refine_$m.json; done
echo "==== two-stage detector (PDV second stage)"
for b in 1 8; do timeout 300 python tools/bench_pdv.py --math f16x2 --batch $b 2>/dev/null | tail -1 > $O/${TAG}_bench_pdv_b$b.json; cat $O/${TAG}_bench_pdv_b$b.json; done
for b in 8 16; do timeout 300 python tools/bench_pdv.py --math f16x2 --batch $b --pipeline 2>/dev/null | tail -1 > $O/${TAG}_bench_pdv_pipeline_b$b.json; cat $O/${TAG}_bench_pdv_pipeline_b$b.json; done
bash tools/gpu_pdv_prof.sh 8 > /dev/null 2>&1; cp gpurun_out/pdv/kernel_trace_pdv_b8.txt $O/${TAG}_kernel_trace_pdv_b8.txt; cp gpurun_out/pdv/phases_b8.txt $O/${TAG}_pdv_phases_b8.txt
fi
find $O -name "*.db" -delete
