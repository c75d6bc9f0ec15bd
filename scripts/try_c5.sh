#!/bin/bash
# timing of script-derived scenes at their own settings
for w in c5 paraffin_lamp; do
  timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || { echo "$w failed"; tail -3 gpurun_out/bench_$w.err; continue; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/bench_$w.json').read().strip().splitlines()[-1])
print('$w', d['value'], d['unit'], 'ms/step', d['ms_per_step'], d.get('stages'))
PY
done
