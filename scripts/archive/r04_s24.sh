#!/bin/bash
# round 4, GPU session 24: reservations of 256 queue slots per atomic instead of 64 (are the queue counters what k_shade_hits waits
# for on many_spheres?): same-box A/B on C3 / 16 and the other workloads; kernel trace of C3 / 16
set -o pipefail
OUT=$PWD/gpurun_out/s24; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
W="--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16"
python bench.py $W --no-cpu-baseline > /dev/null 2>&1
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
O="ACN_LIBDIR=$PWD/lib_q256"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "$W" "$M;$O"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M;$O"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$M;$O"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M;$O"
done
scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$M;$O"
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$M;$O"
echo "ab done" > $OUT/progress.txt
for v in main q256; do
  d=$OUT/trace_$v; mkdir -p $d
  L=$PWD/actinon_amd/lib; [ $v = q256 ] && L=$PWD/lib_q256
  ACN_LIBDIR=$L ACN_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py $W --steps 1 --warmup 0 --no-cpu-baseline > $d/log.txt 2>&1
  cp $(find $d -name "*kernel_stats.csv" | head -1) $OUT/c3_stride16_kernel_stats_$v.csv
  find $d -name "*.csv" -size +5M -delete
done
python3 - <<'PY'
import csv
for v in ("main","q256"):
    print(v)
    for r in list(csv.DictReader(open("gpurun_out/s24/c3_stride16_kernel_stats_%s.csv" % v)))[:8]:
        print("  %-36s %5s %9.1f ms" % (r["Name"][:36], r["Calls"], int(r["TotalDurationNs"])/1e6))
PY
echo session done
