#!/bin/bash
OUT=$PWD/gpurun_out/dbg
mkdir -p $OUT
export TMPDIR=/tmp
for w in smoke c1 c2 c5; do
  echo "== $w" | tee -a $OUT/log.txt
  timeout -k 5 60 python scripts/frame_times.py $w 4 2>&1 | tail -8 | tee -a $OUT/log.txt
  echo "rc $?" | tee -a $OUT/log.txt
done
echo "== pytest" | tee -a $OUT/log.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 100 2>&1 | tail -15 | tee -a $OUT/log.txt
