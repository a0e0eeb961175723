#!/bin/bash
# split-precision engine: parity tests, then fp32 vs split bench (development round)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/split
echo "==== split tests"; timeout 900 python -m pytest tests/test_gpu_split.py -m gpu -q -x --timeout=600 2>&1 | tail -25
