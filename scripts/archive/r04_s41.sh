#!/bin/bash
# Where a cold process' first picture goes, upload included: the upload timeline (stream, events, host tables, device copies, first
# kernel) and the call timeline of ACN_DEBUG_CHUNKS for four workloads, two runs each.   usage: scripts/r04_s41.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for w in wine_glass_1080p c2 paraffin_lamp c5; do
  for v in 1 2; do
    echo "== $w run $v" >> $out/upload_timeline.txt
    ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py $w 2 2>&1 | grep -v "acn chunk" | grep -v "acn sample\] [0-9r]" | cut -c1-260 >> $out/upload_timeline.txt || exit 1
  done
done
echo "== c3 (one frame)" >> $out/upload_timeline.txt
ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py c3 1 2>&1 | grep -v "acn chunk" | grep -v "acn sample\] [0-9r]" | cut -c1-260 >> $out/upload_timeline.txt || exit 1
cat $out/upload_timeline.txt
