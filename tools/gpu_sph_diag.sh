#!/bin/bash
# k_spconv_h<128x128x32> with one effect removed at a time (-DDZ_SPCONV_DIAG build of sparse_conv_h.hip, DZ_TUNE_SPCONV128=11..20 =
# DIAG 1..10 of hgemm_pipeline / the kernel: 1 no MFMAs, 2 no LDS fragment reads, 3 no LDS stage stores, 4 no global loads, 6 / 7 no
# neighbour-index loads, 8 gathers out of range, 9 weight loads out of range, 10 term-major MFMAs; results are garbage, times are not)
cd "$(dirname "$0")/.."
cp detzero_amd/libdetzero_hip.so /tmp/libdz_orig.so
objs=$(ls detzero_amd/csrc/build/*.o | grep -v sparse_conv_h.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDZ_SPCONV_DIAG -c detzero_amd/csrc/sparse_conv_h.hip -o /tmp/sph_diag.o 2>/dev/null || { echo "compile failed"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/sph_diag.o -o detzero_amd/libdetzero_hip.so 2>/dev/null || { echo "link failed"; exit 1; }
LIST=${1:-0 11 12 13 14 16 17 18 19 20}
for d in $LIST; do
  echo "== DZ_TUNE_SPCONV128=$d"
  DZ_TUNE_SPCONV128=$d timeout 200 python tools/bench_spconv.py --batch 16 --math f16x2 --reps 5 2>&1 | grep -E "128->128" | grep -v "+res" | head -1 | cut -c1-30,95-170
done
cp /tmp/libdz_orig.so detzero_amd/libdetzero_hip.so
