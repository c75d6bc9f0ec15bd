#!/bin/bash
# GPU session 29: the instrumented (COUNT = true) k_hard_path spills 132-140 VGPRs, the production variants 612-1058 (same 128-VGPR
# budget; at 168 VGPRs the counts do not change).  Which is faster?  Per-kernel time of both on the configs k_hard_path matters for.
set -o pipefail
OUT=$PWD/gpurun_out/s29
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for cfg in "c4 16" "c5full 256" "c5 1" "paraffin_lamp 1"; do
  set -- $cfg
  for cw in 0 1; do
    if [ $cw = 1 ]; then export ACN_COUNT_WORK=1; else unset ACN_COUNT_WORK; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$1_$cw -o s -- python3 bench.py --workload $1 --pixel-stride $2 --steps 2 --warmup 1 --quick --no-cpu-baseline > $OUT/stats_$1_$cw.log 2>&1 || { tail -n 5 $OUT/stats_$1_$cw.log; exit 1; }
    echo "== $1 count_work=$cw" | tee -a $OUT/summary.txt
    grep '^{' $OUT/stats_$1_$cw.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %.1f ms/step' % d['ms_per_step'])" | tee -a $OUT/summary.txt
    f=$(find $OUT/stats_$1_$cw -name "*kernel_stats.csv" | head -1)
    python3 - $f <<'PY' | tee -a $OUT/summary.txt
import csv, sys, re
for r in list(csv.DictReader(open(sys.argv[1])))[:6]:
    m = re.match(r"(void )?([a-zA-Z_0-9]+(<[^>]*>)?)", r["Name"])
    print("   %-40s calls %5s  total %9.1f ms  avg %8.3f ms  %5.1f %%" % (m.group(2), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
PY
    find $OUT/stats_$1_$cw -name "*.csv" ! -name "*kernel_stats.csv" -delete
  done
done
echo done
