#!/bin/bash
# usage: scripts/ab_stages.sh <outfile> <dir> <bench args> -- VAR=val ... ; ...   (bench.py of <dir>, with per-stage times)
out=$1; dir=$2; args=$3; shift 3
IFS=';' read -ra CFG <<< "$*"
for c in "${CFG[@]}"; do
  line=$(cd $dir && env $c timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('%8.1f Msamples/s %8.2f ms/step | stage sums: walk %.1f shade %.1f hard %.1f | chunks %d retries %d' % (d['value'], d['ms_per_step'], s['walk_ms'], s['shade_ms'], s['hard_ms'], s['chunks'], s['retries']))
")
  echo "[$dir: $c ] $line" | tee -a $out
done
