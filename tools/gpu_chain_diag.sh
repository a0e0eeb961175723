#!/bin/bash
# k_mlp_chain with one effect removed at a time (-DDZ_CHAIN_DIAG=<bits>: results are garbage, times are not).  Runs on the GPU box's
# scratch copy of the repo: the library there is relinked per variant and put back at the end.
#   usage: tools/gpu_chain_diag.sh "0 1 2 8 16 32 64"
cd "$(dirname "$0")/.."
cp detzero_amd/libdetzero_hip.so /tmp/libdz_orig.so
objs=$(ls detzero_amd/csrc/build/*.o | grep -v mlp_chain.o)
for d in $1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDZ_CHAIN_DIAG=$d -c detzero_amd/csrc/mlp_chain.hip -o /tmp/mlp_chain_diag.o 2>/dev/null || { echo "diag $d: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/mlp_chain_diag.o -o detzero_amd/libdetzero_hip.so 2>/dev/null || { echo "diag $d: link failed"; continue; }
  echo "== DZ_CHAIN_DIAG=$d"
  timeout 120 python tools/bench_chain.py 2>/dev/null | grep "memory chain"
done
cp /tmp/libdz_orig.so detzero_amd/libdetzero_hip.so
