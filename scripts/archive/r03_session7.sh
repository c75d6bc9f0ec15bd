#!/bin/bash
# GPU session 7 of round 3: launch geometry on the headline frame (environment only), frame-by-frame times after the workspace fixes
set -o pipefail
OUT=$PWD/gpurun_out/s7
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
B="--workload wine_glass_1080p --steps 10 --warmup 3 --quick"
scripts/ab.sh $OUT/ab.txt "$B" "ACN_X=0;ACN_WALK_GRID=256;ACN_WALK_GRID=256 ACN_SHADE_GRID=1024;ACN_WALK_GRID=256 ACN_SHADE_GRID=1024 ACN_LANES=6;ACN_WALK_GRID=384 ACN_SHADE_GRID=768;ACN_LANES=5;ACN_LANES=6;ACN_LANES=8;ACN_X=1;ACN_PRIVATE_LIMIT=65536;ACN_PRIVATE_LIMIT=131072;ACN_PRIVATE_LIMIT=16384;ACN_FETCH_SHADE=8;ACN_FETCH_SHADE=32;ACN_GRID=768 ACN_SHADE_GRID=768;ACN_SHADE_GRID=768;ACN_SHADE_GRID=1024;ACN_X=2"
echo "ab done" | tee $OUT/progress.txt
for w in c5 paraffin_lamp c2 wine_glass_1080p c1; do
  ( cd $ROOT && timeout -k 10 120 python scripts/frame_times.py $w 10 ) 2>/dev/null | tee -a $OUT/frames.txt
done
echo done | tee -a $OUT/progress.txt
