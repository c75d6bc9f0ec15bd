#!/bin/bash
# GPU session 4 of round 3: same-box A/B of the cone culling and of k_walk at one wave per SIMD; chunking of the heavy configs
set -o pipefail
OUT=gpurun_out/s4
mkdir -p $OUT
export TMPDIR=/tmp
B="--workload wine_glass_1080p --steps 8 --warmup 2 --quick"
echo "== 1080p: cull / no cull / tw1 (twice each, interleaved)" | tee $OUT/progress.txt
scripts/ab.sh $OUT/ab.txt "$B" "ACN_X=0;ACN_LIBDIR=$PWD/lib_nocull;ACN_LIBDIR=$PWD/lib_tw1;ACN_X=1;ACN_LIBDIR=$PWD/lib_nocull ACN_X=1;ACN_LIBDIR=$PWD/lib_tw1 ACN_X=1;ACN_LIBDIR=$PWD/lib_tw1 ACN_WALK_GRID=256"
echo "== stride 8" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab_stride8.txt "$B --pixel-stride 8" "ACN_X=0;ACN_LIBDIR=$PWD/lib_nocull;ACN_LIBDIR=$PWD/lib_tw1;ACN_LIBDIR=$PWD/lib_tw1 ACN_WALK_GRID=256"
echo "== c5 / paraffin" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab_c5.txt "--workload c5 --steps 3 --warmup 1 --quick" "ACN_X=0;ACN_LIBDIR=$PWD/lib_nocull;ACN_LIBDIR=$PWD/lib_tw1"
scripts/ab.sh $OUT/ab_c5.txt "--workload paraffin_lamp --steps 3 --warmup 1 --quick" "ACN_X=0;ACN_LIBDIR=$PWD/lib_nocull;ACN_LIBDIR=$PWD/lib_tw1"
echo "== c3 / c4 every 16th pixel: chunking at the new bound" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab_c34.txt "--workload c3 --steps 1 --warmup 1 --quick --pixel-stride 16" "ACN_X=0;ACN_WORKSPACE_MB=24576;ACN_LIBDIR=$PWD/lib_tw1"
scripts/ab.sh $OUT/ab_c34.txt "--workload c4 --steps 1 --warmup 1 --quick --pixel-stride 16" "ACN_X=0;ACN_WORKSPACE_MB=24576;ACN_LIBDIR=$PWD/lib_tw1"
echo "== c3 full frame" | tee -a $OUT/progress.txt
timeout -k 10 300 python bench.py --workload c3 --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $OUT/checksum_c3.json > $OUT/bench_c3.json 2> $OUT/bench_c3.err; echo "c3 rc $?" | tee -a $OUT/progress.txt
python - <<'PY' | tee -a $OUT/progress.txt
import json
d = json.load(open("gpurun_out/s4/bench_c3.json")); s = d["stages"]
print("c3 %.2f Msamples/s %.0f ms chunks %d retries %d ws %.1f GB sha %s" % (d["value"], d["ms_per_step"], s["chunks"], s["retries"], s["workspace_bytes"] / 1e9, d["frame_check"]["sha256"][:16]))
PY
echo done | tee -a $OUT/progress.txt
