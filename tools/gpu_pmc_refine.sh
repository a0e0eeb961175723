#!/bin/bash
# SQ / LDS counters of the refiner bench per kernel -> gpurun_out/prof/refine_pmc_<math>.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
M=${1:-f16x2}
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/rpmc
( cd /tmp && timeout 400 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof/rpmc -o rf -- python $GRAFT_REPO_ROOT/tools/bench_refine.py --math $M > $GRAFT_REPO_ROOT/gpurun_out/prof/rpmc_stdout.txt 2>&1 )
python tools/rocpd_summary.py gpurun_out/prof/rpmc/rf_results.db > gpurun_out/prof/refine_pmc_$M.txt
find gpurun_out/prof -name "*.db" -delete
grep -E "k_mlp_chain|k_pointnet3|k_mha_block<true>|k_xattn_fold" gpurun_out/prof/refine_pmc_$M.txt | head -40
