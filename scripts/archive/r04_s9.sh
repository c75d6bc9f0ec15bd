#!/bin/bash
# round 4, GPU session 9: cold frames with the tighter sample rates; 5 waves per SIMD for k_hard_shadow / k_hard_path / k_walk;
# lanes and grids for the 1/8 share
set -o pipefail
OUT=$PWD/gpurun_out/s9; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
HS="ACN_LIBDIR=$PWD/lib_hs5"
HP="ACN_LIBDIR=$PWD/lib_hp5"
W5="ACN_LIBDIR=$PWD/lib_w5"
for w in wine_glass_1080p paraffin_lamp c5 c2; do
  echo "== $w" >> $OUT/frames.txt
  ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py $w 4 2> $OUT/chunks_$w.txt | tail -n 4 >> $OUT/frames.txt
done
cut -c1-140 $OUT/frames.txt
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M;$HS;$HP;$W5"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M;$HS;$HP;$W5"
  scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$M;$HS;$HP;$W5"
  scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M;$HS;$HP;$W5"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M"
scripts/ab.sh $OUT/ab_stride8.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$M;$M ACN_LANES=6;$M ACN_LANES=8;$M ACN_GRID=512 ACN_SHADE_GRID=512;$M ACN_GRID=256 ACN_SHADE_GRID=256;$M ACN_LANES=2 ACN_GRID=1024 ACN_SHADE_GRID=1024;$M ACN_PRIVATE_LIMIT=131072;$M ACN_WALK_PASSES=4;$M"
echo session done
