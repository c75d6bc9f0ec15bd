#!/bin/bash
# GPU session 33: max-ILP scheduling of k_walk on the CSG scenes, three interleaved repetitions
set -o pipefail
OUT=$PWD/gpurun_out/s33
mkdir -p $OUT
export TMPDIR=/tmp
A="ACN_LIBDIR=$PWD/actinon_amd/lib"
B="ACN_LIBDIR=$PWD/lib_ilp"
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_c5.txt "--workload c5 --steps 6 --warmup 2 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_paraffin.txt "--workload paraffin_lamp --steps 6 --warmup 2 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_c5full.txt "--workload c5full --steps 2 --warmup 1 --quick --pixel-stride 256" "$A;$B"
done
scripts/ab.sh $OUT/ab_c3.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
echo done
