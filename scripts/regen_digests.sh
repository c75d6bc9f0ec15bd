#!/bin/bash
# Re-renders the frames whose digests tests/golden/frame_checksums.json holds (all but the whole C3 / C4 / C5 frames, which take
# 30 s .. 17 min each: pass "heavy" as the second argument for C3 / C4) and writes one JSON per frame into <outdir>;
# scripts/merge_digests.py folds them into the golden file.   usage (on the GPU box): scripts/regen_digests.sh <outdir> [heavy]
out=$1; mkdir -p $out
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" --steps 1 --warmup 0 --quick --no-cpu-baseline --checksum $out/$name.json > $out/$name.log 2>&1 || { tail -n 5 $out/$name.log; return 1; }; }
run wine_glass_1080p && run c2 --workload c2 && run c1 --workload c1 && run c5 --workload c5 && run paraffin_lamp --workload paraffin_lamp \
  && run c4_stride64 --workload c4 --pixel-stride 64 && run c3_stride64 --workload c3 --pixel-stride 64 || exit 1
if [ "$2" = heavy ]; then run c3 --workload c3 && run c4 --workload c4 || exit 1; fi
echo digests done
