#!/bin/bash
# Lanes made during the upload (ACN_EARLY_LANES=1, the default) against lanes made by the first call (=0): upload + first frames of four
# workloads, three processes each, then the GPU tests that exercise cold handles and lanes.   usage: scripts/r04_s42.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for w in wine_glass_1080p c2 paraffin_lamp c5; do
  for v in 0 1 0 1 0 1; do
    echo "== $w ACN_EARLY_LANES=$v" >> $out/early_lanes.txt
    ACN_EARLY_LANES=$v ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py $w 3 2>&1 | grep -v "acn chunk" | grep -v "acn lane" | grep -v "acn sample\] [0-9r]" | cut -c1-300 >> $out/early_lanes.txt || exit 1
  done
done
grep -E "^==|upload \(|^    " $out/early_lanes.txt | cut -c1-150
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "cold or lane or shard or edge or entry_point or resum" > $out/tests_subset.log 2>&1; tail -n 3 $out/tests_subset.log
