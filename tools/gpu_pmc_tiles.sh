#!/bin/bash
# Counter passes over the eager bench with the tile-resident sparse engine: HBM traffic (FETCH / WRITE, own passes), LDS array
# and bank conflicts, matrix pipe - and the LDS pass of the gather engine for comparison.   -> gpurun_out/r03/tiles_*
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03
mkdir -p $O
A="--steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux --no-refine --no-pdv"
pass() {   # tag, counters, extra bench args
  local tag=$1 c=$2; shift 2
  rm -rf $O/p_$tag; ( cd /tmp && timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/p_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py $A "$@" > $O/p_${tag}_stdout.txt 2>&1 )
  python tools/rocpd_summary.py $O/p_$tag/b_results.db | sed -n '/PMC/,$p' | grep -E "k_spconv" > $O/${tag}.txt; head -30 $O/${tag}.txt
}
pass tiles_pmc_FETCH_SIZE FETCH_SIZE --sparse-engine tiles
pass tiles_pmc_WRITE_SIZE WRITE_SIZE --sparse-engine tiles
pass tiles_pmc_LDS "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS" --sparse-engine tiles
pass gather_pmc_LDS "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
find $O -name "*.db" -delete
