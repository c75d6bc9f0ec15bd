#!/bin/bash
# GPU session 18: learned launch grids, on / off, same box, interleaved
set -o pipefail
OUT=$PWD/gpurun_out/s18
mkdir -p $OUT
export TMPDIR=/tmp
echo "== gpu tests" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.txt
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1"
  scripts/ab.sh $OUT/ab_stride8.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1"
  scripts/ab.sh $OUT/ab_c2.txt "--workload c2 --steps 10 --warmup 3 --quick" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c1 --steps 20 --warmup 3 --quick" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1"
scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1"
scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1"
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "ACN_LEARN_GRIDS=0;ACN_LEARN_GRIDS=1"
echo done | tee -a $OUT/progress.txt
