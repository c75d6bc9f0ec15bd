#!/bin/bash
# HBM traffic of the heavy configs on a pixel lattice (two separate PMC passes each: FETCH_SIZE, WRITE_SIZE), per kernel and per frame.
# usage: scripts/pmc_traffic_heavy.sh <outdir under gpurun_out/>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for cfg in "c3 16" "c4 16" "c5full 256"; do
  set -- $cfg; w=$1; s=$2
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --quick --no-cpu-baseline --pixel-stride $s > $out/bench_${w}_stride$s.json 2>/dev/null || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$out/pmc_${w}_$c; mkdir -p $d
    rocprofv3 --pmc $c --output-format csv -d $d -o t -- python3 bench.py --workload $w --steps 2 --warmup 0 --quick --no-cpu-baseline --pixel-stride $s > $d/log.txt 2>&1 || { tail -n 5 $d/log.txt; exit 1; }
  done
  python3 - "$out" "$w" "$s" <<'PY' >> $out/traffic_heavy.txt
import csv, sys, json, os, re, collections, glob
out, w, s = sys.argv[1], sys.argv[2], sys.argv[3]
passes = 2
per = collections.defaultdict(lambda: [0.0, 0.0])
for i, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    f = glob.glob(os.path.join(out, "pmc_%s_%s" % (w, c), "**", "*counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and r["Kernel_Name"].startswith(("void k_", "k_")):
            m = re.match(r"(void )?([a-zA-Z_0-9]+)(<[^>]*>)?", r["Kernel_Name"])
            per[m.group(2) + (m.group(3) or "")][i] += float(r["Counter_Value"])
d = json.loads([l for l in open(os.path.join(out, "bench_%s_stride%s.json" % (w, s))) if l.startswith("{")][-1])
ms = d["ms_per_step"]
tot = sum(f * 2 + wr for f, wr in per.values()) * 1024 / passes
print("== %s, every %sth pixel: %.0f ms per frame (unprofiled run), %.1f GB per frame (FETCH_SIZE x 2 + WRITE_SIZE, learning pass of the cold handle included) = %.0f GB/s = %.1f %% of 8 TB/s" % (w, s, ms, tot / 1e9, tot / 1e9 / (ms * 1e-3), tot / 1e9 / (ms * 1e-3) / 80.0))
for k, (f, wr) in sorted(per.items(), key=lambda x: -(x[1][0] * 2 + x[1][1]))[:8]:
    print("   %-40s fetch(x2) %8.2f GB  write %8.2f GB per frame" % (k, f * 2 * 1024 / passes / 1e9, wr * 1024 / passes / 1e9))
PY
  rm -rf $out/pmc_${w}_FETCH_SIZE $out/pmc_${w}_WRITE_SIZE
done
cat $out/traffic_heavy.txt
