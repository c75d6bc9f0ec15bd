#!/bin/bash
# round 4, GPU session 10: walk passes / private limit across workloads (the 1/8 share gained 11 % with ACN_WALK_PASSES=4)
set -o pipefail
OUT=$PWD/gpurun_out/s10; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
V="$M;$M ACN_WALK_PASSES=3;$M ACN_WALK_PASSES=4;$M ACN_WALK_PASSES=6;$M ACN_PRIVATE_LIMIT=131072;$M ACN_PRIVATE_LIMIT=65536 ACN_WALK_PASSES=4"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_stride8.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$V"
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$V"
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$V"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$V"
scripts/ab.sh $OUT/ab_other.txt "--workload c1 --steps 20 --warmup 3 --quick" "$V"
echo session done
