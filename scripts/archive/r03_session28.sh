#!/bin/bash
# GPU session 28: per-kernel time of the heavy configs with this round's closing code (rocprofv3 --kernel-trace --stats)
set -o pipefail
OUT=$PWD/gpurun_out/s28
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for cfg in "c3 16" "c4 16" "c5full 256"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$1 -o s -- python3 bench.py --workload $1 --pixel-stride $2 --steps 2 --warmup 1 --quick --no-cpu-baseline > $OUT/stats_$1.log 2>&1 || { tail -n 5 $OUT/stats_$1.log; exit 1; }
  grep '^{' $OUT/stats_$1.log | cut -c1-160
  find $OUT/stats_$1 -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
echo done
