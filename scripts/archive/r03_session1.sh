#!/bin/bash
# GPU session 1 of round 3: parity suite on the new host code, then A/B of kernel variants and launch geometry on the headline frame.
set -o pipefail
mkdir -p gpurun_out/s1
OUT=gpurun_out/s1
export TMPDIR=/tmp
echo "== gpu tests (parity + configs)" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.txt
B="--workload wine_glass_1080p --steps 8 --warmup 2 --quick"
echo "== baseline" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab.txt "$B" "ACN_X=0"
echo "== shade waves 3" | tee -a $OUT/progress.txt
[ -d lib_sw3 ] && scripts/ab.sh $OUT/ab.txt "$B" "ACN_LIBDIR=$PWD/lib_sw3"
echo "== geometry" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab.txt "$B" "ACN_WALK_GRID=256;ACN_WALK_GRID=256 ACN_SHADE_GRID=1024;ACN_WALK_GRID=256 ACN_LANES=6;ACN_LANES=6;ACN_LANES=8;ACN_WALK_GRID=256 ACN_LANES=8;ACN_WALK_GRID=384;ACN_LANES=2;ACN_LANES=3"
echo "== workspace" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab.txt "$B" "ACN_WORKSPACE_MB=4096;ACN_WORKSPACE_MB=16384;ACN_WORKSPACE_MB=65536"
echo "== phase ticks" | tee -a $OUT/progress.txt
[ -d lib_pt ] && ACN_LIBDIR=$PWD/lib_pt timeout -k 10 300 python scripts/phase_ticks.py wine_glass_1080p > $OUT/phase_ticks_wine_glass.txt 2>&1
echo "== full bench line" | tee -a $OUT/progress.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 2 --checksum $OUT/checksum_wine_glass_1080p.json > $OUT/bench_wine_glass_1080p.json 2> $OUT/bench.err; echo "bench rc $?" | tee -a $OUT/progress.txt
echo "== pixel stride 8 (share of one of 8 ranks)" | tee -a $OUT/progress.txt
scripts/ab.sh $OUT/ab_stride8.txt "--workload wine_glass_1080p --steps 8 --warmup 2 --quick --pixel-stride 8" "ACN_X=0;ACN_LANES=2;ACN_LANES=1"
echo done | tee -a $OUT/progress.txt
