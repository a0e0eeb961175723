#!/bin/bash
# per-layer sparse conv timing for tile-variant knobs (development)
cd "$(dirname "$0")/.."
for v in ${VARIANTS:-"0,0" "1,1"}; do
  a=${v%,*}; b=${v#*,}
  echo "== SPCONV64=$a SPCONV128=$b"
  DZ_TUNE_SPCONV64=$a DZ_TUNE_SPCONV128=$b timeout 200 python tools/bench_spconv.py --batch ${BATCH:-4} --math f16x2 2>&1 | grep "^k\|^sum" | grep -v "+res" | sed 's/taps.*128\] *[0-9. ]*  *\([0-9.]* us\)/\1/' | uniq
done
