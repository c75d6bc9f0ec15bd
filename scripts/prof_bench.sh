#!/bin/bash
# rocprofv3 kernel trace + stats of a bench.py run; summaries land in gpurun_out/prof_<tag>/
# usage: scripts/prof_bench.sh <tag> [bench args...]
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python3 bench.py --no-cpu-baseline "$@" > $out/bench.log 2>&1
ls -R $out | head -30
