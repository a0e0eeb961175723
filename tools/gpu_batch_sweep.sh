#!/bin/bash
# dense-stage grouping: its test, then the batch sweep of the headline (frames per pass 8 .. 32).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -x --timeout=300 -p no:cacheprovider -k "dense_stage_in_frame_groups or batched_frames" 2>&1 | tail -3 | tee gpurun_out/sweep_tests.txt
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["config"].get("frames_per_step_per_gpu"), d["value"], d["ms_per_step"])'
for b in 8 16 24 32; do
  timeout 400 python bench.py --batch $b --steps 60 --warmup 5 --no-cpu-baseline --no-aux --no-refine --no-pdv --profile-frames 0 2>gpurun_out/sweep_$b.err | tail -1 | python -c "$P" || tail -3 gpurun_out/sweep_$b.err
done | tee gpurun_out/sweep.txt
