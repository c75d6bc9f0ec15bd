#!/bin/bash
# HBM traffic of one bench run: two separate PMC passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).
# usage: scripts/pmc_traffic.sh <workload> <steps>
set -e
w=$1; steps=$2
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/pmc_$c
  mkdir -p $d
  rocprofv3 --pmc $c --output-format csv -d $d -o t -- python3 bench.py --workload $w --steps $steps --warmup 0 --no-cpu-baseline > $d/log.txt 2>&1
done
python3 - "$w" "$steps" <<'PY'
import csv, sys, json, os
w, steps = sys.argv[1], int(sys.argv[2])
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{c}/t_counter_collection.csv")))
    tot[c] = sum(float(r["Counter_Value"]) for r in rows if r["Counter_Name"] == c and r["Kernel_Name"].startswith(("void k_", "k_")))
# bench executes steps + min(2, steps) recorded + 1 instrumented pass
passes = steps + min(2, max(1, steps)) + 1
fetch_b = tot["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads
write_b = tot["WRITE_SIZE"] * 1024
sys.path.insert(0, os.getcwd())
import bench
out = {"workload": w, "passes": passes, "kernel_source_hash": bench.kernel_source_hash(), "FETCH_SIZE_KB_raw": tot["FETCH_SIZE"], "WRITE_SIZE_KB_raw": tot["WRITE_SIZE"],
       "hbm_bytes_per_step": (fetch_b + write_b) / passes, "fetch_bytes_per_step_x2": fetch_b / passes, "write_bytes_per_step": write_b / passes}
print(json.dumps(out))
json.dump(out, open("gpurun_out/traffic_%s.json" % w, "w"))
PY
