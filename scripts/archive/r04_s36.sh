#!/bin/bash
# A/B of the staggered first call (ACN_COLD_PIPELINE): first / second / third frame of a cold handle, interleaved twice, and the
# tests that render on cold handles.       usage: scripts/r04_s36.sh <outdir under gpurun_out/>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "cold or digest or whole_frame or overflow or shard" > $out/tests_subset.log 2>&1 || { tail -n 20 $out/tests_subset.log; exit 1; }
tail -n 1 $out/tests_subset.log
for rep in 1 2; do
  for w in wine_glass_1080p c2 paraffin_lamp c5; do
    for v in 0 1; do
      echo "== ACN_COLD_PIPELINE=$v $w (repetition $rep)" >> $out/frames_cold_pipeline.txt
      ACN_COLD_PIPELINE=$v timeout -k 10 200 python scripts/frame_times.py $w 4 2>&1 | tail -n 4 >> $out/frames_cold_pipeline.txt || exit 1
    done
  done
done
cut -c1-120 $out/frames_cold_pipeline.txt
