#!/bin/bash
# whole C5 frame, second half of the bands, then one band again with the chunk log (ACN_DEBUG_CHUNKS) to see why chunks are redone
bash scripts/r04_c5bands.sh 1620 1755 1890 2025 2160
OUT=$PWD/gpurun_out/c5full_r04
ACN_DEBUG_CHUNKS=1 timeout -k 10 400 python bench.py --workload c5full --rows 1350:1485 --steps 1 --warmup 0 --quick --no-cpu-baseline > $OUT/debug_rows_1350_1485.json 2> $OUT/debug_rows_1350_1485.err
grep -c "acn chunk" $OUT/debug_rows_1350_1485.err
