#!/bin/bash
# One band of BASELINE.json configs[4] at its stated size (hanging_lamp 3840x2160, path_samples 1024, direct_samples 30) on one GPU.
# usage: scripts/r03_c5band.sh Y0 Y1 [extra bench args]
set -o pipefail
Y0=$1; Y1=$2; shift 2
OUT=$PWD/gpurun_out/c5full
mkdir -p $OUT
export TMPDIR=/tmp
( while true; do sleep 60; echo "alive $(date +%T)" >> $OUT/alive_$Y0.txt; done ) &
BEAT=$!
timeout -k 10 1080 python bench.py --workload c5full --rows $Y0:$Y1 --steps 1 --warmup 0 --quick --checksum $OUT/checksum_rows_${Y0}_$Y1.json "$@" > $OUT/bench_rows_${Y0}_$Y1.json 2> $OUT/bench_rows_${Y0}_$Y1.err
rc=$?
kill $BEAT
echo "rc $rc" | tee $OUT/rc_$Y0.txt
python - <<PY
import json
d = json.load(open("$OUT/bench_rows_${Y0}_$Y1.json")); s = d["stages"]
print("c5full rows $Y0:$Y1  %.3f Msamples/s  %.1f s  chunks %d retries %d ws %.1f GB sha %s" % (d["value"], d["ms_per_step"] / 1e3, s["chunks"], s["retries"], s["workspace_bytes"] / 1e9, d["frame_check"]["sha256"][:16]))
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:200], "speedup", d["speedup_vs_cpu_baseline"])
PY
