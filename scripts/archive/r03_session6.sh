#!/bin/bash
# GPU session 6 of round 3: parity suite, then frame-by-frame times, round-2 tree against the current one
set -o pipefail
OUT=$PWD/gpurun_out/s6
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
echo "== gpu tests" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.txt
for w in c5 paraffin_lamp c2 wine_glass_1080p; do
  for rep in 1 2; do
    ( cd $ROOT/old_r2 && timeout -k 10 120 python $ROOT/scripts/frame_times.py $w 8 ) 2>/dev/null | tee -a $OUT/frames.txt
    ( cd $ROOT && timeout -k 10 120 python scripts/frame_times.py $w 8 ) 2>/dev/null | tee -a $OUT/frames.txt
    ( cd $ROOT && ACN_WS_UNIFORM=1 timeout -k 10 120 python scripts/frame_times.py $w 8 ) 2>/dev/null | sed 's/ repo$/ repo ACN_WS_UNIFORM=1/' | tee -a $OUT/frames.txt
  done
  echo "$w done" | tee -a $OUT/progress.txt
done
echo done | tee -a $OUT/progress.txt
