#!/bin/bash
# k_build_neighbors_rows with one effect removed at a time (-DDZ_NBR_DIAG=n build of sparse_index.hip: 1 no bitmap / prefix loads,
# 2 no table stores, 3 neither; the tables are garbage, the times are not): the pyramid time of tools/bench_index.py
cd "$(dirname "$0")/.."
cp detzero_amd/libdetzero_hip.so /tmp/libdz_orig.so
objs=$(ls detzero_amd/csrc/build/*.o | grep -v sparse_index.o)
for d in ${1:-0 1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -DDZ_NBR_DIAG=$d -c detzero_amd/csrc/sparse_index.hip -o /tmp/si_diag.o 2>/dev/null || { echo "compile failed"; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/si_diag.o -o detzero_amd/libdetzero_hip.so 2>/dev/null || { echo "link failed"; exit 1; }
  echo "== DZ_NBR_DIAG=$d"; timeout 300 python tools/bench_index.py 2>/dev/null | tail -1 | cut -c1-120
done
cp /tmp/libdz_orig.so detzero_amd/libdetzero_hip.so
