#!/bin/bash
# GPU session 32: smaller persistent grids for k_shade and the hard-ray kernels (scratch footprint of the resident waves against L2 / MALL)
set -o pipefail
OUT=$PWD/gpurun_out/s32
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
V="ACN_X=0;ACN_SHADE_GRID=512;ACN_SHADE_GRID=384;ACN_SHADE_GRID=256;ACN_GRID=512;ACN_GRID=256;ACN_SHADE_GRID=512 ACN_GRID=512"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$V"
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
echo done
