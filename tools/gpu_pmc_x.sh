#!/bin/bash
# SQ counter passes over the 32 / 64 / 128-channel submanifold layers of the per-layer bench, x-run engine and gather engine
# usage: tools/gpu_pmc_x.sh [layers, default 32-32,64-64,128-128]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ONLY=${1:-32-32,64-64,128-128}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  for E in xrun gather; do
    rm -rf $OUT/pmcx_${E}_$i
    ( cd /tmp && DZ_TUNE_SPCONV_ENGINE=$E timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/pmcx_${E}_$i -o sp -- python $GRAFT_REPO_ROOT/tools/bench_spconv.py --batch 16 --math f16x2 --reps 2 --only $ONLY > $OUT/pmcx_${E}_${i}_stdout.txt 2>&1 )
    echo "== $E pass $i"
    python tools/rocpd_summary.py $OUT/pmcx_${E}_$i/sp_results.db --json $OUT/pmcx_${E}_$i.json | sed -n '/PMC/,$p' | grep -E "k_spconv" | cut -c1-40,80-200 | head -40
  done
done
find $OUT -name "*.db" -delete
