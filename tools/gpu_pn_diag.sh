#!/bin/bash
# k_pointnet3 with one effect removed at a time (-DDZ_PN_DIAG=<bits>: results are garbage, times are not); see tools/gpu_chain_diag.sh
cd "$(dirname "$0")/.."
cp detzero_amd/libdetzero_hip.so /tmp/libdz_orig.so
objs=$(ls detzero_amd/csrc/build/*.o | grep -v pointnet.o)
for d in $1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDZ_PN_DIAG=$d -c detzero_amd/csrc/pointnet.hip -o /tmp/pn_diag.o 2>/dev/null || { echo "diag $d: compile failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/pn_diag.o -o detzero_amd/libdetzero_hip.so 2>/dev/null || { echo "diag $d: link failed"; continue; }
  echo "== DZ_PN_DIAG=$d"
  timeout 120 python tools/bench_chain.py 2>/dev/null | grep "encoder"
done
cp /tmp/libdz_orig.so detzero_amd/libdetzero_hip.so
