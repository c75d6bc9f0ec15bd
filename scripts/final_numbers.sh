#!/bin/bash
# the numbers quoted in DESIGN.md 7: every BASELINE.json config once
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_1080p.json 2> gpurun_out/final/bench_1080p.err && tail -c 600 gpurun_out/final/bench_1080p.json | head -c 0
for w in c2 c1 c5 paraffin_lamp; do
  timeout -k 10 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_$w.json 2> gpurun_out/final/bench_$w.err || echo "$w failed"
done
timeout -k 10 300 python scripts/time_scene.py diamond 480 270 512 50 > gpurun_out/final/diamond_480.txt 2>&1
timeout -k 10 300 python scripts/time_scene.py many_spheres:5:0 480 270 256 20 > gpurun_out/final/many_spheres_480.txt 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/final/bench_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], '%.1f Msamples/s  %.1f ms/step' % (d['value'], d['ms_per_step']), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e:
        print(f, 'ERR', e)
PY
grep "iter 1" gpurun_out/final/*.txt
