#!/bin/bash
# Where a cold handle's first frame goes: the call timeline of ACN_DEBUG_CHUNKS (lanes made, learning pass, queues sized, lanes done),
# with the lanes made beside the learning pass (ACN_COLD_PIPELINE=1, the default) and before it (=0).   usage: scripts/r04_s38.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for w in wine_glass_1080p c2 paraffin_lamp c5; do
  for v in 0 1 0 1; do
    echo "== $w ACN_COLD_PIPELINE=$v" >> $out/first_frame_timeline.txt
    ACN_COLD_PIPELINE=$v ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py $w 3 2>&1 | grep -v "acn chunk" | grep -v "acn sample\] [0-9r]" | cut -c1-220 >> $out/first_frame_timeline.txt || exit 1
  done
done
cat $out/first_frame_timeline.txt
