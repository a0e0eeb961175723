#!/bin/bash
# k_spconv_x with one effect removed at a time (-DDZ_SPCONV_DIAG build of sparse_conv_x.hip, DZ_TUNE_X_DIAG bits: 1 no MFMAs, 2 no
# fragment LDS reads, 4 no weight loads, 8 no window loads, 16 no barriers, 32 no epilogue, 64 row addresses built once per tile;
# results are garbage, times are not).   tools/gpu_x_diag.sh ["0 1 2 ..."]
cd "$(dirname "$0")/.."
cp detzero_amd/libdetzero_hip.so /tmp/libdz_orig.so
objs=$(ls detzero_amd/csrc/build/*.o | grep -v sparse_conv_x.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-inline-asm -DDZ_SPCONV_DIAG -c detzero_amd/csrc/sparse_conv_x.hip -o /tmp/spx_diag.o 2>/dev/null || { echo "compile failed"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/spx_diag.o -o detzero_amd/libdetzero_hip.so 2>/dev/null || { echo "link failed"; exit 1; }
LIST=${1:-0 1 2 3 4 8 12 15 16 32 64}
for d in $LIST; do
  echo "== DZ_TUNE_X_DIAG=$d"
  DZ_TUNE_SPCONV_ENGINE=xrun DZ_TUNE_X_DIAG=$d timeout 200 python tools/bench_spconv.py --batch 16 --math f16x2 --reps 5 --only 32-32,64-64,128-128 > /tmp/xd.txt 2>&1; grep -E "^x" /tmp/xd.txt | grep -v "+res" | sort -u | cut -c1-30,95-125; grep x-dbg /tmp/xd.txt | sort -u
done
cp /tmp/libdz_orig.so detzero_amd/libdetzero_hip.so
