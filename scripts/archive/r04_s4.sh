#!/bin/bash
# round 4, GPU session 4: k_walk at 4 waves per SIMD as the default; k_shade<16> at 5 / 3 waves; lanes and grids on the new balance
set -o pipefail
OUT=$PWD/gpurun_out/s4; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
S5="ACN_LIBDIR=$PWD/lib_shade5"
S3="ACN_LIBDIR=$PWD/lib_shade3"
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for rep in 1 2 3; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M;$S5;$S3"
done
scripts/ab.sh $OUT/ab_knobs.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M ACN_LANES=2;$M ACN_LANES=3;$M ACN_LANES=6;$M ACN_GRID=768;$M ACN_GRID=1024;$M ACN_SHADE_GRID=768;$M ACN_SHADE_GRID=1024;$M ACN_WALK_GRID=384;$M ACN_PRIVATE_LIMIT=16384;$M ACN_PRIVATE_LIMIT=65536;$M ACN_FETCH_WALK=128;$M"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_other.txt "--workload c2 --steps 10 --warmup 3 --quick" "$M;$S5"
  scripts/ab.sh $OUT/ab_other.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M;$S5"
  scripts/ab.sh $OUT/ab_other.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$M;$M ACN_LANES=2;$M ACN_LANES=3"
done
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o s -- python3 bench.py --steps 5 --warmup 2 --quick --no-cpu-baseline > $OUT/stats1.log 2>&1
find $OUT/stats1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
for pass in a b; do
  if [ $pass = a ]; then C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS"; else C="SQ_WAVE_CYCLES SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; fi
  d=$OUT/pmc_$pass; mkdir -p $d
  ACN_LANES=1 rocprofv3 --pmc $C --output-format csv -d $d -o t -- python3 bench.py --steps 1 --warmup 0 --quick --no-cpu-baseline > $d/log.txt 2>&1 && python3 scripts/pmc_summary.py $(find $d -name "*counter_collection.csv" | head -1) > $OUT/pmc_$pass.txt
  find $d -name "*.csv" -size +5M -delete
done
echo session done
