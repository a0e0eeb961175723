#!/bin/bash
# LDS-array occupancy of the conv kernels inside the detector: SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT against SQ_BUSY_CU_CYCLES
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/lds; mkdir -p $O; rm -rf $O/p
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace -d $O/p -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --profile-frames 0 --no-aux > $O/stdout.txt 2>&1 )
python tools/rocpd_summary.py $O/p/b_results.db | grep -E "k_spconv|k_conv3x3|k_conv2d" | grep -E "LDS|BUSY_CU" 
find $O -name "*.db" -delete
