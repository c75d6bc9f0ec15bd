#!/bin/bash
# round 4, GPU session 12: defaults after the walk-pass change and the trim window: suite, frames, the workloads
set -o pipefail
OUT=$PWD/gpurun_out/s12; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for w in wine_glass_1080p paraffin_lamp c5 c2; do
  echo "== $w" >> $OUT/frames.txt
  timeout -k 10 200 python scripts/frame_times.py $w 14 2>/dev/null | tail -n 14 >> $OUT/frames.txt
done
cut -c1-120 $OUT/frames.txt
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$M"
  scripts/ab.sh $OUT/ab.txt "--workload c2 --steps 10 --warmup 3 --quick" "$M"
  scripts/ab.sh $OUT/ab.txt "--workload c1 --steps 20 --warmup 3 --quick" "$M"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M"
  scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$M"
done
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M"
scripts/ab.sh $OUT/ab.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M"
echo session done
