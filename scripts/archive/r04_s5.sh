#!/bin/bash
# round 4, GPU session 5: chunk_alloc without per-lane constants (the wave's reservation state through a scalar pointer, ranks by mbcnt)
set -o pipefail
OUT=$PWD/gpurun_out/s5; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
S4="ACN_LIBDIR=$PWD/lib_s4"
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$S4;$M"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$S4;$M"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$S4;$M"
  scripts/ab.sh $OUT/ab_other.txt "--workload c5 --steps 4 --warmup 2 --quick" "$S4;$M"
  scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$S4;$M"
done
scripts/ab.sh $OUT/ab_other.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$S4;$M"
scripts/ab.sh $OUT/ab_other.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$S4;$M"
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $OUT/stats1.log 2>&1
find $OUT/stats1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
for pass in a; do
  C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS"
  d=$OUT/pmc_$pass; mkdir -p $d
  ACN_LANES=1 rocprofv3 --pmc $C --output-format csv -d $d -o t -- python3 bench.py --steps 1 --warmup 0 --quick --no-cpu-baseline > $d/log.txt 2>&1 && python3 scripts/pmc_summary.py $(find $d -name "*counter_collection.csv" | head -1) > $OUT/pmc_$pass.txt
  find $d -name "*.csv" -size +5M -delete
done
echo session done
