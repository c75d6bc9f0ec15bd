#!/bin/bash
# C5 at stated size (3840x2160 p1024): CPU baseline on a lattice of pixels spread over the WHOLE raster (no centred window), beside the GPU on
# every 256th pixel.   usage: scripts/r04_s45.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python bench.py --workload c5full --steps 1 --warmup 1 --quick --pixel-stride 256 > $out/bench_c5full_stride256_cpu_whole_frame_lattice.json 2> $out/bench_c5full.err || { tail -n 5 $out/bench_c5full.err; exit 1; }
cut -c1-200 $out/bench_c5full_stride256_cpu_whole_frame_lattice.json
echo s45 done
