#!/bin/bash
# GPU session 30: digests of further whole frames for tests/golden/frame_checksums.json (twice each: the digest must repeat)
set -o pipefail
OUT=$PWD/gpurun_out/s30
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for a in "--workload c2" "--workload c1" "--workload c5" "--workload paraffin_lamp" "--workload c4 --pixel-stride 64" "--workload c3 --pixel-stride 64"; do
  i=$((i+1))
  for rep in a b; do
    ACN_LANES=$([ $rep = a ] && echo 4 || echo 2) timeout -k 10 300 python bench.py $a --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/digest_${i}_$rep.json > $OUT/line_${i}_$rep.json 2> $OUT/err_${i}_$rep.txt || { tail -n 3 $OUT/err_${i}_$rep.txt; exit 1; }
  done
  python3 - $OUT/digest_${i}_a.json $OUT/digest_${i}_b.json <<'PY'
import json, sys
a, b = (json.load(open(f)) for f in sys.argv[1:3])
for k in a:
    print(k, a[k]["sha256"][:16], "repeats with 2 lanes" if b[k]["sha256"] == a[k]["sha256"] else "DIFFERS")
PY
done
echo done
