#!/bin/bash
# GPU session 27: HBM traffic of the heavy configs (every 16th / 16th / 256th pixel, two production passes each: FETCH_SIZE and
# WRITE_SIZE in separate rocprofv3 --pmc passes), the whole C3 / C4 frames on a WARM handle, and the headline line once more
# (bench.py now takes roofline.kernel_ms from the last timed step)
set -o pipefail
OUT=$PWD/gpurun_out/s27
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for cfg in "c3 16" "c4 16" "c5full 256"; do
  set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/pmc_$1_$c; mkdir -p $d
    rocprofv3 --pmc $c --output-format csv -d $d -o t -- python3 bench.py --workload $1 --pixel-stride $2 --steps 1 --warmup 1 --quick --no-cpu-baseline > $d/log.txt 2>&1 || { tail -n 5 $d/log.txt; exit 1; }
    grep '^{' $d/log.txt | cut -c1-150
  done
  python3 - $OUT $1 $2 <<'PY' | tee -a $OUT/traffic_heavy.txt
import csv, sys, os, glob, json
out, w, stride = sys.argv[1], sys.argv[2], sys.argv[3]
passes = 2
tot = {}; ms = None
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, "pmc_%s_%s" % (w, c), "**", "*counter_collection.csv"), recursive=True)[0]
    tot[c] = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and r["Kernel_Name"].startswith(("void k_", "k_")))
fetch_b = tot["FETCH_SIZE"] * 1024 * 2 / passes    # gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads
write_b = tot["WRITE_SIZE"] * 1024 / passes
print("%-7s every %3sth pixel: fetch(x2) %8.2f GB  write %8.2f GB  total %8.2f GB per pass (two passes, the first on a cold handle)" % (w, stride, fetch_b / 1e9, write_b / 1e9, (fetch_b + write_b) / 1e9))
PY
  find $OUT/pmc_$1_FETCH_SIZE $OUT/pmc_$1_WRITE_SIZE -name "*.csv" -delete
done
for w in c4 c3; do
  timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 1 --quick --no-cpu-baseline > $OUT/bench_${w}_warm.json 2> $OUT/bench_${w}_warm.err || { tail -n 3 $OUT/bench_${w}_warm.err; exit 1; }
  cut -c1-170 $OUT/bench_${w}_warm.json
done
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --checksum $OUT/checksum_wine_glass_1080p.json > $OUT/bench_wine_glass_1080p.json 2> $OUT/bench.err || { tail -n 5 $OUT/bench.err; exit 1; }
cut -c1-200 $OUT/bench_wine_glass_1080p.json
echo done
