#!/bin/bash
# PMC passes over the per-layer sparse conv bench (development): HBM-side bytes, L2 hit rate, L1 requests, wave stall split.
# usage: tools/gpu_pmc_spconv.sh [tag]   (env: DZ_TUNE_* knobs are passed through)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-sp}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u > $OUT/counters_avail.txt
wc -l $OUT/counters_avail.txt
i=0
for C in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf $OUT/pmc_${TAG}_$i
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmc_${TAG}_$i -o sp -- python $GRAFT_REPO_ROOT/tools/bench_spconv.py --batch 16 --math f16x2 --reps 2 > $OUT/pmc_${TAG}_${i}_stdout.txt 2>&1 )
  echo "== pass $i: $C"; tail -2 $OUT/pmc_${TAG}_${i}_stdout.txt | cut -c1-200
  python tools/rocpd_summary.py $OUT/pmc_${TAG}_$i/sp_results.db --json $OUT/pmc_${TAG}_$i.json | sed -n '/PMC/,$p' | grep -E "k_spconv" | head -60
done
find $OUT -name "*.db" -delete
