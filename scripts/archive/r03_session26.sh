#!/bin/bash
# GPU session 26: GPU suite with the script-made texture / distance scene; C1 and C2 against the private-stack limit
set -o pipefail
OUT=$PWD/gpurun_out/s26
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 400 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt
tail -3 $OUT/pytest_gpu.log | tee -a $OUT/progress.txt
python bench.py --workload wine_glass_1080p --steps 3 --warmup 2 --quick --no-cpu-baseline > /dev/null 2>&1
V="ACN_X=0;ACN_PRIVATE_LIMIT=131072;ACN_PRIVATE_LIMIT=1048576;ACN_WALK_PASSES=3;ACN_WALK_PASSES=2"
for rep in 1 2; do
  scripts/ab.sh $OUT/c1.txt "--workload c1 --steps 30 --warmup 5 --quick" "$V"
  scripts/ab.sh $OUT/c2.txt "--workload c2 --steps 10 --warmup 3 --quick" "$V"
  scripts/ab.sh $OUT/p1080.txt "--workload wine_glass_1080p --steps 10 --warmup 3 --quick" "$V"
done
echo done
