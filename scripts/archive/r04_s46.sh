#!/bin/bash
# one rank's share of the 1080p frame under both splits, world 1 / 2 / 4 / 8, alone on one GPU.   usage: scripts/r04_s46.sh <outdir>
out=gpurun_out/$1; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python scripts/share_probe.py 8 > $out/share_probe.txt 2> $out/share_probe.err || { tail -n 8 $out/share_probe.err; exit 1; }
cat $out/share_probe.txt
