#!/bin/bash
# per-kernel times, 1080p wine_glass, one lane: round-2 tree against the current one
OUT=$PWD/gpurun_out/s15
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd $ROOT && timeout -k 10 200 python bench.py --steps 6 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2; do
( cd $ROOT/old_r2 && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02_$rep -o s -- python3 bench.py --steps 6 --warmup 2 --quick --no-cpu-baseline > $OUT/r02_$rep.log 2>&1 )
( cd $ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/now_$rep -o s -- python3 bench.py --steps 6 --warmup 2 --quick --no-cpu-baseline > $OUT/now_$rep.log 2>&1 )
done
for d in r02_1 now_1 r02_2 now_2; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d" | tee -a $OUT/kernel_stats.txt
  python3 - "$f" <<'PY' | tee -a $OUT/kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("  %-60s calls %5s total %9.1f ms avg %8.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  find $OUT/$d -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
done
grep '^{' $OUT/r02_1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('r02 ms', d['ms_per_step'])"
grep '^{' $OUT/now_1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('now ms', d['ms_per_step'])"
