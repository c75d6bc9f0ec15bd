#!/bin/bash
# round 4, GPU session 25: k_shade_hits reserves queue slots in pieces that follow its input (64 ... 512 per atomic): suite, same-box
# A/B against fixed 64 and against a ceiling of 1024, kernel trace of C3 / 16
set -o pipefail
OUT=$PWD/gpurun_out/s25; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; rc=$?; echo "pytest rc $rc" | tee $OUT/progress.txt; tail -n 8 $OUT/tests_gpu.log
[ $rc -eq 0 ] || exit 1
W="--workload c3 --steps 2 --warmup 1 --quick --pixel-stride 16"
python bench.py $W --no-cpu-baseline > /dev/null 2>&1
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
O="ACN_LIBDIR=$PWD/lib_qfix"
K="ACN_LIBDIR=$PWD/lib_q1024"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "$W" "$O;$M;$K"
done
for rep in 1 2; do
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$O;$M;$K"
  scripts/ab.sh $OUT/ab.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick --pixel-stride 8" "$O;$M;$K"
  scripts/ab.sh $OUT/ab.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$O;$M;$K"
  scripts/ab.sh $OUT/ab.txt "--workload c2 --steps 10 --warmup 3 --quick" "$O;$M;$K"
done
scripts/ab.sh $OUT/ab.txt "--workload c5 --steps 4 --warmup 2 --quick" "$O;$M;$K"
scripts/ab.sh $OUT/ab.txt "--workload c4 --steps 2 --warmup 1 --quick --pixel-stride 16" "$O;$M;$K"
echo "ab done" >> $OUT/progress.txt
d=$OUT/trace_main; mkdir -p $d
ACN_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 bench.py $W --steps 1 --warmup 1 --no-cpu-baseline > $d/log.txt 2>&1
cp $(find $d -name "*kernel_stats.csv" | head -1) $OUT/c3_stride16_kernel_stats.csv
find $d -name "*.csv" -size +5M -delete
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/s25/c3_stride16_kernel_stats.csv")))[:8]:
    print("  %-36s %5s %9.1f ms" % (r["Name"][:36], r["Calls"], int(r["TotalDurationNs"])/1e6))
PY
echo session done
