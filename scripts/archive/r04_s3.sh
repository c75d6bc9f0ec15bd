#!/bin/bash
# round 4, GPU session 3: looped pair code (one copy of the operand code per "one operand, then the other"), one copy of the root
# traversal for lights + matter; occupancy variants of k_walk / k_hard_path on the smaller kernels
set -o pipefail
OUT=$PWD/gpurun_out/s3; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
S2="ACN_LIBDIR=$PWD/lib_s2"
L="ACN_LIBDIR=$PWD/lib_loops"
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
W3="ACN_LIBDIR=$PWD/lib_walk3"
W4="ACN_LIBDIR=$PWD/lib_walk4"
W4G="ACN_LIBDIR=$PWD/lib_walk4 ACN_WALK_GRID=1024"
H2="ACN_LIBDIR=$PWD/lib_hpath2"
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$S2;$L;$M;$W3;$W4;$W4G;$H2"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$S2;$M;$W3;$W4"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$S2;$L;$M;$W3;$W4;$H2"
  scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$S2;$L;$M;$W3;$W4;$H2"
  scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$S2;$M;$W3;$W4"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$S2;$M;$W3;$W4;$H2"
scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$S2;$M"
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $OUT/stats1.log 2>&1
find $OUT/stats1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo session done
