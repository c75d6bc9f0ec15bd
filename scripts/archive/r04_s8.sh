#!/bin/bash
# round 4, GPU session 8: cold handles with dead-slot allowance; C3 with the prefetching table walk; kernel shares of C4 / C5;
# HBM traffic of the headline frame
set -o pipefail
OUT=$PWD/gpurun_out/s8; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
N="ACN_LIBDIR=$PWD/lib_nopf"
for w in wine_glass_1080p paraffin_lamp c5 c2; do
  echo "== $w" >> $OUT/frames.txt
  ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py $w 5 2> $OUT/chunks_$w.txt | tail -n 5 >> $OUT/frames.txt
done
cut -c1-140 $OUT/frames.txt
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_1080p.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M"
  scripts/ab.sh $OUT/ab_c3.txt "--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16" "$N;$M"
done
for w in c4 c5full; do
  st=16; [ $w = c5full ] && st=256
  ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -o s -- python3 bench.py --workload $w --pixel-stride $st --steps 1 --warmup 1 --quick --no-cpu-baseline > $OUT/stats_$w.log 2>&1
  find $OUT/stats_$w -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -o s -- python3 bench.py --workload c5 --steps 3 --warmup 1 --quick --no-cpu-baseline > $OUT/stats_c5.log 2>&1
find $OUT/stats_c5 -name "*.csv" ! -name "*kernel_stats.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/pmc_$c; mkdir -p $d
  rocprofv3 --pmc $c --output-format csv -d $d -o t -- python3 bench.py --steps 4 --warmup 0 --quick --no-cpu-baseline > $d/log.txt 2>&1
  python3 scripts/pmc_summary.py $(find $d -name "*counter_collection.csv" | head -1) > $OUT/pmc_$c.txt
  find $d -name "*.csv" -size +5M -delete
done
echo session done
