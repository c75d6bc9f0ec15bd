#!/bin/bash
# round 4, GPU session 6: cold handles -- the strided learning pass (learn_rates) against round 3's behaviour (ACN_LEARN_SAMPLE=0);
# ACN_LDS_MAX=0 (no staged nodes: root-level leaves read their nodes through scalar loads)
set -o pipefail
OUT=$PWD/gpurun_out/s6; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -n 3 $OUT/tests_gpu.log
python bench.py --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1   # warm the box
for w in wine_glass_1080p paraffin_lamp c5 c2; do
  for v in 1 0; do
    echo "== $w ACN_LEARN_SAMPLE=$v" >> $OUT/frames.txt
    ACN_LEARN_SAMPLE=$v ACN_DEBUG_CHUNKS=1 timeout -k 10 200 python scripts/frame_times.py $w 5 2> $OUT/chunks_${w}_$v.txt | tail -n 5 >> $OUT/frames.txt
  done
done
cut -c1-140 $OUT/frames.txt
M="ACN_LIBDIR=$PWD/actinon_amd/lib"
for rep in 1 2; do
  scripts/ab.sh $OUT/ab_lds.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$M;$M ACN_LDS_MAX=0"
  scripts/ab.sh $OUT/ab_lds.txt "--workload paraffin_lamp --steps 4 --warmup 2 --quick" "$M;$M ACN_LDS_MAX=0"
done
# cold whole C3 frame, then a warm one on the same handle (bench: warmup 0 / 1)
for v in 1 0; do
  ACN_LEARN_SAMPLE=$v ACN_DEBUG_CHUNKS=1 timeout -k 10 300 python bench.py --workload c3 --steps 1 --warmup 0 --quick --no-cpu-baseline 2> $OUT/chunks_c3_cold_$v.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('c3 cold LEARN_SAMPLE=$v %8.1f ms  chunks %d retries %d' % (d['ms_per_step'], s['chunks'], s['retries']))" | tee -a $OUT/c3_cold.txt
done
timeout -k 10 300 python bench.py --workload c3 --steps 1 --warmup 1 --quick --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('c3 warm %8.1f ms  chunks %d retries %d' % (d['ms_per_step'], s['chunks'], s['retries']))" | tee -a $OUT/c3_cold.txt
echo session done
