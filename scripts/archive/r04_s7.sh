#!/bin/bash
# round 4, GPU session 7: k_shade fission (direct-light half + deferred shadow rays on a side stream), exact sample rates for cold
# handles, nodes staged in LDS only for scenes with generic nested compounds
set -o pipefail
OUT=$PWD/gpurun_out/s7; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not committed_digest" > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M ACN_SHADE_FISSION=0;$M;$M ACN_LDS_MAX=40960"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$M ACN_SHADE_FISSION=0;$M"
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$M ACN_SHADE_FISSION=0;$M"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M ACN_SHADE_FISSION=0;$M"
  scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$M ACN_SHADE_FISSION=0;$M"
  scripts/ab.sh $OUT/ab_other.txt "--workload c1 --steps 20 --warmup 3 --quick" "$M ACN_SHADE_FISSION=0;$M"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M ACN_SHADE_FISSION=0;$M;$M ACN_LDS_MAX=40960"
scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M ACN_SHADE_FISSION=0;$M"
for w in wine_glass_1080p paraffin_lamp c5 c2; do
  echo "== $w" >> $OUT/frames.txt
  ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py $w 5 2> $OUT/chunks_$w.txt | tail -n 5 >> $OUT/frames.txt
done
cut -c1-140 $OUT/frames.txt
scripts/regen_digests.sh $OUT/digests | tee -a $OUT/progress.txt
echo session done
