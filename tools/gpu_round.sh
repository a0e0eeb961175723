#!/bin/bash
# The evidence set of a round on ONE GPU box (replaces gpu_round2/3/4.sh, gpu_profiles.sh, gpu_pmc*.sh, gpu_trace*.sh, gpu_x_diag.sh,
# gpu_pdv_prof.sh of earlier rounds): gpu test-suite, smoke, the default bench line (all legs), rocprofv3 kernel trace of the headline
# command, HBM-side PMC passes (FETCH_SIZE / WRITE_SIZE, separate passes - never together with a trace domain) and the SQ matrix-pipe
# counters of the same command, the per-layer sparse benchmark of both engines, the x-run kernel's in-kernel cycle accounting
# (diag build), the refiner and two-stage (PDV) benches with their traces.
# usage: tools/gpu_round.sh <tag> [sections]     tag = r05a ...; sections = any of: tests bench trace pmc sq clk layers xdiag refine pdv
#        (default: all).  Outputs -> gpurun_out/<round>/<tag>_*  (copy what is to be judged into profiles/).
TAG=${1:-r05a}
SECT=${2:-tests bench trace pmc sq clk layers xdiag refine pdv}
cd "$(dirname "$0")/.."
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:0:3}
mkdir -p $O
has() { case " $SECT " in *" $1 "*) return 0;; *) return 1;; esac; }
COMMON="--steps 20 --warmup 5 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux --no-refine --no-pdv"
PARGS="--steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux --no-refine --no-pdv"
trace() {   # name, command...
  local name=$1; shift
  echo "==== rocprofv3 kernel-trace: $name ($*)"
  rm -rf $O/trace_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_$name -o t -- "$@" > $O/trace_${name}_stdout.txt 2>&1 )
  python tools/rocpd_summary.py $O/trace_$name/t_results.db > $O/${TAG}_kernel_trace_$name.txt; head -${TRACE_HEAD:-16} $O/${TAG}_kernel_trace_$name.txt
}
if has tests; then
  echo "==== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider 2>&1 | tail -4 | tee $O/${TAG}_gputests.txt
  # the achieved errors the parity tests print (per stage and math mode, boxes, the float64 error budget, the pre-scale runs, the two-stage pins)
  timeout 900 python -m pytest tests/test_gpu_full_parity.py tests/test_gpu_split.py tests/test_gpu_f16.py tests/test_pdv.py tests/test_gpu_sequence_chain.py -m gpu -q -s --timeout=600 -p no:cacheprovider 2>&1 |
    grep -E "parity |boxes |max \|error\||^  stage|^  x_conv|^  encoded|^  spatial|^  head/|worst f16x2|^gain|gain [0-9.e+-]+:|two-stage at|two_stage vs|full-size|f16 \(single|f16x2 on the same|sequence anchor|passed|failed" | cut -c1-400 > $O/${TAG}_gputests_parity.txt
  tail -3 $O/${TAG}_gputests_parity.txt
  echo "==== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 | tee $O/${TAG}_smoke.txt
fi
if has bench; then
  echo "==== bench (default command)"; timeout 900 python bench.py 2> $O/${TAG}_bench.err > $O/${TAG}_bench_graph.json; tail -14 $O/${TAG}_bench.err
fi
if has trace; then trace bench_eager20 python $ROOT/bench.py $COMMON; fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "==== rocprofv3 pmc $c"
    rm -rf $O/pmc_$c; ( cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o bench -- python $ROOT/bench.py $PARGS > $O/pmc_${c}_stdout.txt 2>&1 )
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
    if 'spconv' in k or 'conv3x3' in k or 'conv2d' in k:
        rd = 2.0 * 1024.0 * v.get('FETCH_SIZE', {'per_call': 0})['per_call']; wr = 1024.0 * v.get('WRITE_SIZE', {'per_call': 0})['per_call']
        print('%-60s read %7.1f MB  write %7.1f MB per launch' % (k[:60], rd / 1e6, wr / 1e6))
PY
fi
if has sq; then
  echo "==== rocprofv3 pmc SQ (matrix pipe)"
  rm -rf $O/pmc_sq; ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_sq -o bench -- python $ROOT/bench.py $PARGS > $O/pmc_sq_stdout.txt 2>&1 )
  python tools/rocpd_summary.py $O/pmc_sq/bench_results.db | sed -n '/PMC/,$p' > $O/${TAG}_pmc_SQ_bench_eager3.txt; grep -E "MFMA_BUSY" $O/${TAG}_pmc_SQ_bench_eager3.txt | head -10
fi
if has clk; then
  # idle CUs or clock (round-5 review, item 4): GRBM_GUI_ACTIVE next to SQ_BUSY_CU_CYCLES in ONE pass of the same eager command, and an
  # amd-smi / rocm-smi clock sample taken while the un-profiled bench runs
  echo "==== rocprofv3 pmc GRBM_GUI_ACTIVE + SQ_BUSY_CU_CYCLES (effective clock, busy CUs)"
  rm -rf $O/pmc_clk; ( cd /tmp && timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/pmc_clk -o bench -- python $ROOT/bench.py $PARGS > $O/pmc_clk_stdout.txt 2>&1 )
  python tools/clock_table.py $O/pmc_clk/bench_results.db > $O/${TAG}_clock_table.txt 2>&1; head -16 $O/${TAG}_clock_table.txt
  ( timeout 60 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-frames 0 --no-aux --no-refine --no-pdv > /dev/null 2>&1 & 
    sleep 25; for i in 1 2 3 4 5; do (rocm-smi --showclocks 2>/dev/null | grep -i -E "sclk|mclk|fclk" | head -3; rocm-smi --showpower 2>/dev/null | grep -i -E "power" | head -2) ; sleep 1; done; wait ) > $O/${TAG}_smi_clock_sample.txt 2>&1
  tail -12 $O/${TAG}_smi_clock_sample.txt
fi
if has layers; then
  echo "==== per-layer sparse benchmark, both engines"
  for e in gather xrun; do DZ_TUNE_SPCONV_ENGINE=$e timeout 300 python tools/bench_spconv.py --batch 16 --reps 20 --math f16x2 2>&1 | tail -23 > $O/${TAG}_spconv_layers_$e.txt; tail -1 $O/${TAG}_spconv_layers_$e.txt; done
fi
if has xdiag; then
  # k_spconv_x with in-kernel cycle counters / one effect removed at a time (-DDZ_SPCONV_DIAG build of sparse_conv_x.hip; DZ_TUNE_X_DIAG
  # bits: 1 no MFMAs, 2 no fragment LDS reads, 4 no weight loads, 8 no window loads, 16 no barriers, 32 no epilogue, 512 cycle counters;
  # results are garbage for all but 512, times are not).  XDIAG="512 1 2 ..." picks the list.
  echo "==== x-run kernel: cycle accounting (diag build)"
  cp detzero_amd/libdetzero_hip.so /tmp/libdz_orig.so
  objs=$(ls detzero_amd/csrc/build/*.o | grep -v sparse_conv_x.o)
  if /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DDZ_SPCONV_DIAG -c detzero_amd/csrc/sparse_conv_x.hip -o /tmp/spx_diag.o 2>/dev/null &&
     /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/spx_diag.o -o detzero_amd/libdetzero_hip.so 2>/dev/null; then
    for d in ${XDIAG:-512}; do
      echo "== DZ_TUNE_X_DIAG=$d"
      DZ_TUNE_SPCONV_ENGINE=xrun DZ_TUNE_X_DIAG=$d timeout 200 python tools/bench_spconv.py --batch 16 --math f16x2 --reps 5 --only 32-32,64-64,128-128 > /tmp/xd.txt 2>&1
      grep -E "^x" /tmp/xd.txt | grep -v "+res" | sort -u | cut -c1-30,95-125; grep x-dbg /tmp/xd.txt | sort -u
    done 2>&1 | grep -v "steps/wave *per wave" | tee $O/${TAG}_xrun_cycles.txt
  else echo "diag build failed"; fi
  cp /tmp/libdz_orig.so detzero_amd/libdetzero_hip.so
fi
if has refine; then
  echo "==== refiner"
  for m in f32 f16x2; do timeout 300 python tools/bench_refine.py --math $m 2>/dev/null | tail -1 > $O/${TAG}_bench_refine_$m.json; cut -c1-500 $O/${TAG}_bench_refine_$m.json; done
  TRACE_HEAD=36 trace refine_f16x2 python $ROOT/tools/bench_refine.py --math f16x2 --steps 2
fi
if has pdv; then
  echo "==== two-stage detector (PDV second stage)"
  for b in 1 8; do timeout 300 python tools/bench_pdv.py --math f16x2 --batch $b 2>/dev/null | tail -1 > $O/${TAG}_bench_pdv_b$b.json; cat $O/${TAG}_bench_pdv_b$b.json; done
  for b in 8 16; do timeout 300 python tools/bench_pdv.py --math f16x2 --batch $b --pipeline 2>/dev/null | tail -1 > $O/${TAG}_bench_pdv_pipeline_b$b.json; cat $O/${TAG}_bench_pdv_pipeline_b$b.json; done
  timeout 300 python tools/bench_pdv.py --math f16x2 --batch 8 --reps 5 --phases > $O/${TAG}_pdv_phases_b8.json 2> $O/${TAG}_pdv_phases_b8.txt; tail -20 $O/${TAG}_pdv_phases_b8.txt
  TRACE_HEAD=45 trace pdv_b8 python $ROOT/tools/bench_pdv.py --math f16x2 --batch 8 --reps 5 ${PDV_TRACE_ARGS}
fi
find $O -name "*.db" -delete
