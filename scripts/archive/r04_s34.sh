#!/bin/bash
# round 4, GPU session 34: a cold handle at path_samples 1024 gets its learning pass (it had been skipped: the guess allowed 63 positions):
# two bands of the hanging_lamp frame at stated size again, with the chunk log
OUT=$PWD/gpurun_out/s34; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { echo "SMOKE FAILED"; tail -n 15 $OUT/smoke.log; exit 1; }
for r in 1350:1485 2025:2160; do
  ACN_DEBUG_CHUNKS=1 timeout -k 10 400 python bench.py --workload c5full --rows $r --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/checksum_rows_${r/:/_}.json > $OUT/bench_rows_${r/:/_}.json 2> $OUT/debug_rows_${r/:/_}.err
  python3 - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_rows_${r/:/_}.json") if l.startswith("{")][-1]); s = d["stages"]
print("c5full rows $r  %.3f Msamples/s  %.1f s  chunks %d retries %d sha %s" % (d["value"], d["ms_per_step"] / 1e3, s["chunks"], s["retries"], d["frame_check"]["sha256"][:16]))
PY
  grep "acn sample" $OUT/debug_rows_${r/:/_}.err | cut -c1-160
  grep -c OVERFLOW $OUT/debug_rows_${r/:/_}.err
done
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "cold or overflow or config" > $OUT/tests.log 2>&1; echo "pytest rc $?"; tail -n 3 $OUT/tests.log
echo session done
