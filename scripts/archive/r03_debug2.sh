#!/bin/bash
OUT=$PWD/gpurun_out/dbg2
mkdir -p $OUT
export TMPDIR=/tmp
ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python bench.py --workload c5full --steps 1 --warmup 1 --quick --pixel-stride 256 --no-cpu-baseline > $OUT/c5full_s256.json 2> $OUT/c5full_s256.err
grep "acn chunk" $OUT/c5full_s256.err | cut -c1-400
( cd old_r2 && timeout -k 10 200 python bench.py --workload c5full --steps 1 --warmup 1 --quick --pixel-stride 256 --no-cpu-baseline ) 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('r02', d['ms_per_step'], d['stages'])"
