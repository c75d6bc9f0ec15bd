#!/bin/bash
# round 4, GPU session 15: broad phase of the root loops (root_candidates) and the new lane defaults: suite + A/B
set -o pipefail
OUT=$PWD/gpurun_out/s15; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
N="ACN_LIBDIR=$PWD/lib_nocand"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$N;$M"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$N;$M"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$N;$M"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$N;$M"
  scripts/ab.sh $OUT/ab.txt "--workload c2 --steps 10 --warmup 3 --quick" "$N;$M"
done
scripts/ab.sh $OUT/ab.txt "--workload c5full --steps 1 --warmup 1 --quick --pixel-stride 256" "$N;$M"
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$N;$M"
scripts/ab.sh $OUT/ab.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$N;$M"
scripts/ab.sh $OUT/ab.txt "--workload c1 --steps 20 --warmup 3 --quick" "$N;$M"
echo session done
