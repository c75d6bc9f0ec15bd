#!/bin/bash
# GPU session 23: learned grids sized for one workgroup-load per workgroup (ACN_GRID_PASSES 1, 2; 8 ~ the grids before; off)
set -o pipefail
OUT=$PWD/gpurun_out/s23
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
V="ACN_LEARN_GRIDS=0;ACN_GRID_PASSES=1;ACN_GRID_PASSES=2;ACN_GRID_PASSES=8"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_paraffin.txt "--workload paraffin_lamp --steps 5 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab_c5.txt "--workload c5 --steps 5 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab_stride8.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$V"
  scripts/ab.sh $OUT/ab_c2.txt "--workload c2 --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab_c1.txt "--workload c1 --steps 20 --warmup 3 --quick" "$V"
done
scripts/ab.sh $OUT/ab_heavy.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
scripts/ab.sh $OUT/ab_heavy.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
echo done
