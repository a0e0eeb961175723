#!/bin/bash
# Run every -m gpu test FUNCTION in its own process (a faulting kernel poisons the HIP context of the
# whole process), with a per-function timeout; logs and a summary go to gpurun_out/.
# usage: tools/run_gpu_tests.sh [pytest -k expression]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/tests
mkdir -p $OUT
rm -f $OUT/summary.txt
python -m pytest tests -m gpu --collect-only -q ${1:+-k "$1"} 2>/dev/null | grep '::' | sed 's/\[.*//' | sort -u > $OUT/ids.txt
echo "collected $(wc -l < $OUT/ids.txt) test functions"
while read -r id; do
  name=$(echo "$id" | tr '/:' '__')
  start=$(date +%s)
  timeout 600 python -m pytest "$id" -q -x --timeout=300 -p no:cacheprovider > "$OUT/$name.log" 2>&1
  rc=$?
  end=$(date +%s)
  res=$(tail -1 "$OUT/$name.log")
  echo "rc=$rc t=$((end-start))s $id :: $res" | tee -a $OUT/summary.txt
done < $OUT/ids.txt
echo "==== failures ====" | tee -a $OUT/summary.txt
grep -v "^rc=0 " $OUT/summary.txt | grep "^rc=" | tee -a $OUT/failures.txt
for f in $(grep -v "^rc=0 " $OUT/summary.txt | grep "^rc=" | awk '{print $3}' | tr '/:' '__'); do
  echo "---- $f"; grep -E "^E |Error|error|assert" "$OUT/$f.log" | head -25
done
exit 0
