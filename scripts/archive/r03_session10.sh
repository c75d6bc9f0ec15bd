#!/bin/bash
# GPU session 10: per-kernel times on hanging_lamp 600x800 and paraffin_lamp, round-2 tree / current / current without prefetch
set -o pipefail
OUT=$PWD/gpurun_out/s10
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd $ROOT && timeout -k 10 200 python bench.py --steps 6 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
for w in c5 paraffin_lamp; do
  ( cd $ROOT/old_r2 && ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02_$w -o s -- python3 bench.py --workload $w --steps 4 --warmup 2 --quick --no-cpu-baseline > $OUT/r02_$w.log 2>&1 )
  ( cd $ROOT && ACN_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/now_$w -o s -- python3 bench.py --workload $w --steps 4 --warmup 2 --quick --no-cpu-baseline > $OUT/now_$w.log 2>&1 )
  ( cd $ROOT && ACN_LANES=1 ACN_LIBDIR=$ROOT/lib_nopf rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nopf_$w -o s -- python3 bench.py --workload $w --steps 4 --warmup 2 --quick --no-cpu-baseline > $OUT/nopf_$w.log 2>&1 )
  echo "$w done" | tee -a $OUT/progress.txt
done
for d in r02_c5 now_c5 nopf_c5 r02_paraffin_lamp now_paraffin_lamp nopf_paraffin_lamp; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d" | tee -a $OUT/kernel_stats.txt
  python3 - "$f" <<'PY' | tee -a $OUT/kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("  %-60s calls %5s total %9.1f ms avg %8.3f ms" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  find $OUT/$d -name "*.csv" ! -name "*kernel_stats.csv" -delete 2>/dev/null
done
echo done | tee -a $OUT/progress.txt
