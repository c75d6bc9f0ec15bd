#!/bin/bash
# round 4, GPU session 11: why the default walk passes were 68.6 ms in session 10 and 51.8 in session 9 (same binary): per-frame times
set -o pipefail
OUT=$PWD/gpurun_out/s11; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2; do
  for v in "" "ACN_WALK_PASSES=4" "ACN_LEARN_PASSES=0"; do
    echo "== [$v] rep $rep" >> $OUT/frames.txt
    env $v timeout -k 10 200 python scripts/frame_times.py wine_glass_1080p 14 2>/dev/null | tail -n 14 >> $OUT/frames.txt
  done
done
cut -c1-120 $OUT/frames.txt
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M;$M ACN_WALK_PASSES=4;$M;$M ACN_WALK_PASSES=4;$M ACN_LEARN_PASSES=0"
echo session done
