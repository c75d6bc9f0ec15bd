#!/bin/bash
# GPU session 22: why paraffin_lamp (400x600 p30) is slower and less steady than with the round-2 host code
set -o pipefail
OUT=$PWD/gpurun_out/s22
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
ft() {  # dir label workload env...
  local dir=$1 label=$2 w=$3; shift 3
  echo "== $label $*" | tee -a $OUT/frames.txt
  ( cd $dir && env "$@" timeout -k 10 200 python $ROOT/scripts/frame_times.py $w 8 2>&1 | tail -n 9 ) | tee -a $OUT/frames.txt
}
for w in paraffin_lamp c5; do
  ft $ROOT/old_r2 r02 $w ACN_X=0
  ft $ROOT now $w ACN_X=0
  ft $ROOT now $w ACN_LEARN_PASSES=0
  ft $ROOT now $w ACN_LEARN_GRIDS=0
  ft $ROOT now $w ACN_LEARN_PASSES=0 ACN_LEARN_GRIDS=0
  ft $ROOT now $w ACN_WS_UNIFORM=1
  ft $ROOT now $w ACN_LEARN_PASSES=0 ACN_LEARN_GRIDS=0 ACN_WS_UNIFORM=1
  ft $ROOT/old_r2 r02 $w ACN_X=0
done
ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py paraffin_lamp 4 > $OUT/paraffin_chunks.txt 2>&1
echo done
