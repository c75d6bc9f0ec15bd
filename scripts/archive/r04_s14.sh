#!/bin/bash
# round 4, GPU session 14: six lanes on grids of one workgroup per CU (the 1/8 share: 12.2 -> 11.3 ms) across the workloads
set -o pipefail
OUT=$PWD/gpurun_out/s14; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
V="$M;$M ACN_LANES=6 ACN_GRID=256 ACN_SHADE_GRID=256;$M ACN_LANES=6 ACN_GRID=256 ACN_SHADE_GRID=384;$M ACN_LANES=5 ACN_GRID=320 ACN_SHADE_GRID=320;$M ACN_LANES=8 ACN_GRID=192 ACN_SHADE_GRID=192"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 2" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 4" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload c2 --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$V"
done
scripts/ab.sh $OUT/ab.txt "--workload c1 --steps 20 --warmup 3 --quick" "$M;$M ACN_GRID=256 ACN_SHADE_GRID=256"
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M;$M ACN_GRID=256 ACN_SHADE_GRID=256"
scripts/ab.sh $OUT/ab.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M;$M ACN_GRID=256 ACN_SHADE_GRID=256"
echo session done
