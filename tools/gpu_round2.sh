#!/bin/bash
# Round-2 evidence set on one GPU box: gpu test-suite, smoke, the bench line (all legs), rocprofv3 kernel trace + HBM-side PMC
# passes of the same command (separate passes), SQ matrix-pipe counters, and the refiner / TTA benches with their own trace.
# usage: tools/gpu_round2.sh [tag]     outputs -> gpurun_out/r02/<tag>_*
TAG=${1:-r02c}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02
mkdir -p $O
echo "==== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -4 | tee $O/${TAG}_gputests.txt
echo "==== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/${TAG}_smoke.txt
echo "==== bench"; timeout 600 python bench.py --steps 200 --warmup 10 2> $O/${TAG}_bench.err > $O/${TAG}_bench_graph.json; tail -9 $O/${TAG}_bench.err
BARGS="--steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux"
echo "==== rocprofv3 kernel-trace"
rm -rf $O/trace; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py $BARGS > $O/trace_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/trace/bench_results.db > $O/${TAG}_kernel_trace_bench_eager20.txt; head -24 $O/${TAG}_kernel_trace_bench_eager20.txt
PARGS="--steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux"
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
timeout 300 python tools/bench_refine.py 2>/dev/null | tail -1 > $O/${TAG}_bench_refine_f32.json; cut -c1-400 $O/${TAG}_bench_refine_f32.json
timeout 300 python tools/bench_refine.py --math f16x2 2>/dev/null | tail -1 > $O/${TAG}_bench_refine_f16x2.json; cut -c1-400 $O/${TAG}_bench_refine_f16x2.json
rm -rf $O/trace_refine; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_refine -o refine -- python $GRAFT_REPO_ROOT/tools/bench_refine.py --objects 256 --steps 2 > $O/trace_refine_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/trace_refine/refine_results.db > $O/${TAG}_kernel_trace_refine.txt; head -14 $O/${TAG}_kernel_trace_refine.txt
rm -rf $O/pmc_refine; ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace -d $O/pmc_refine -o refine -- python $GRAFT_REPO_ROOT/tools/bench_refine.py --objects 256 --steps 2 > $O/pmc_refine_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/pmc_refine/refine_results.db | sed -n '/PMC/,$p' > $O/${TAG}_pmc_SQ_refine.txt; grep -E "k_mha|k_linear" $O/${TAG}_pmc_SQ_refine.txt | head -12
echo "==== matrix-pipe micro-benchmarks (built by: hipcc --offload-arch=gfx950 -O3 tools/micro/<name>.hip -o tools/micro/<name>)"
for b in mfma_peak mfma_lds; do [ -x tools/micro/$b ] && ./tools/micro/$b > $O/${TAG}_micro_$b.txt 2>&1 && cat $O/${TAG}_micro_$b.txt; done
echo "==== conv3x3 alone: operand data vs time (power)"
for d in randn relu zero; do timeout 120 python tools/bench_conv3x3.py --data $d 2>&1 | tail -1; done | tee $O/${TAG}_conv3x3_data.txt
echo "==== two-stage detector (PDV second stage)"
timeout 300 python tools/bench_pdv.py 2>/dev/null | tail -1 > $O/${TAG}_bench_pdv.json; cat $O/${TAG}_bench_pdv.json
rm -rf $O/trace_pdv; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_pdv -o pdv -- python $GRAFT_REPO_ROOT/tools/bench_pdv.py --reps 5 > $O/trace_pdv_stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/trace_pdv/pdv_results.db > $O/${TAG}_kernel_trace_pdv.txt; head -12 $O/${TAG}_kernel_trace_pdv.txt
echo "==== tta"
timeout 300 python tools/bench_tta.py 2>/dev/null | tail -1 > $O/${TAG}_bench_tta.json; cut -c1-300 $O/${TAG}_bench_tta.json
find $O -name "*.db" -delete
