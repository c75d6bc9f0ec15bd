#!/bin/bash
# (1) whose is the first stream of a cold process (scripts/first_stream.py);  (2) C3 / C4 whole frames with the CPU baseline on a lattice of
# pixels spread over the WHOLE raster (no centred window: an unbiased estimate of the frame's CPU time).   usage: scripts/r04_s44.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
for v in 0 1 0 1; do
  timeout -k 10 120 python scripts/first_stream.py $v c2 2>&1 | grep -E "host stream|host alloc|acn upload|upload:" >> $out/first_stream.txt || exit 1
done
cat $out/first_stream.txt | cut -c1-250
for w in c3 c4; do
  timeout -k 10 500 python bench.py --workload $w --steps 1 --warmup 0 --quick --checksum $out/checksum_$w.json > $out/bench_${w}_cpu_whole_frame_lattice.json 2> $out/bench_$w.err || { tail -n 5 $out/bench_$w.err; exit 1; }
  cut -c1-160 $out/bench_${w}_cpu_whole_frame_lattice.json
done
echo s44 done
