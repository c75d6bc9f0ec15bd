#!/bin/bash
# round 4, GPU session 1: parity suite on the new kernels (closed-form direct Oren-Nayar, k_shade task frames in LDS), same-box
# A/B against the round-3 build (old_r3/), new digests, one-lane kernel trace
set -o pipefail
OUT=$PWD/gpurun_out/s1; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not committed_digest" > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
A="ACN_LIBDIR=$PWD/old_r3/actinon_amd/lib"
B="ACN_LIBDIR=$PWD/actinon_amd/lib"
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$A;$B"
done
scripts/regen_digests.sh $OUT/digests | tee -a $OUT/progress.txt
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$A;$B"
  scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$A;$B"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$A;$B"
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $OUT/stats1.log 2>&1
find $OUT/stats1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo session done
