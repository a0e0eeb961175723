#!/bin/bash
# PDV second stage: its tests (+ the dynamic-VFE tests of the kernel suite), then the 8-frame profile (tools/gpu_pdv_prof.sh) and the
# FramePipeline.two_stage timing.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/pdv
timeout 900 python -m pytest tests/test_pdv.py -q -x --timeout=300 -p no:cacheprovider -m gpu -s 2>&1 | grep -v "^$" | tail -40 | tee gpurun_out/pdv/tests.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x --timeout=300 -p no:cacheprovider -k "dyn or vfe or batched_frames or multisweep" 2>&1 | tail -3 | tee -a gpurun_out/pdv/tests.txt
if [ "$1" != "noprof" ]; then bash tools/gpu_pdv_prof.sh 8; fi
for b in 8 16; do timeout 300 python tools/bench_pdv.py --math f16x2 --batch $b --reps 5 --pipeline 2>&1 | tail -2 | tee gpurun_out/pdv/pipeline_b$b.json; done
