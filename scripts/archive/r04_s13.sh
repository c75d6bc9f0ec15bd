#!/bin/bash
# round 4, GPU session 13: grids and lanes for the 1/8 share (12.2 ms; 11.5 wanted) -- and what they cost the whole frame
set -o pipefail
OUT=$PWD/gpurun_out/s13; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
V="$M;$M ACN_WALK_PASSES=3;$M ACN_GRID=256 ACN_SHADE_GRID=256;$M ACN_GRID=384 ACN_SHADE_GRID=384;$M ACN_GRID=256 ACN_SHADE_GRID=512;$M ACN_GRID=256 ACN_SHADE_GRID=256 ACN_WALK_PASSES=3;$M ACN_LANES=3 ACN_GRID=384 ACN_SHADE_GRID=384;$M ACN_LANES=6 ACN_GRID=256 ACN_SHADE_GRID=256;$M ACN_FETCH_SHADE=4;$M ACN_FETCH_HARD=64;$M ACN_GRID=128 ACN_SHADE_GRID=128"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_stride8.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$V"
done
scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M;$M ACN_GRID=256 ACN_SHADE_GRID=256;$M ACN_GRID=384 ACN_SHADE_GRID=384"
scripts/ab.sh $OUT/ab_other.txt "--workload c1 --steps 20 --warmup 3 --quick" "$M;$M ACN_GRID=256 ACN_SHADE_GRID=256;$M ACN_GRID=128 ACN_SHADE_GRID=128"
echo session done
