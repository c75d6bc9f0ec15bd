#!/bin/bash
# round 4, GPU session 2: escape fix (normals / hit objects no longer forced into scratch by the two real calls), roughness as a
# real call (k_walk 309k -> 152k instructions), camera ray / ray record re-read after the traversal: parity suite + same-box A/B
set -o pipefail
OUT=$PWD/gpurun_out/s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
S1="ACN_LIBDIR=$PWD/old_s1/actinon_amd/lib"
V1="ACN_LIBDIR=$PWD/lib_roughinline"
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
R="ACN_LIBDIR=$PWD/lib_refetch"
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$S1;$V1;$M;$R"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$S1;$V1;$M;$R"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$S1;$V1;$M;$R"
  scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$S1;$V1;$M;$R"
  scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$S1;$M;$R"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$S1;$V1;$M"
scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$S1;$M"
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $OUT/stats1.log 2>&1
find $OUT/stats1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
echo session done
