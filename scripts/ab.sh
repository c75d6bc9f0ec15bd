#!/bin/bash
# usage: scripts/ab.sh <outfile> <bench args> -- VAR=val ... ; VAR=val ... ; ...   (one bench.py run per ';'-separated environment)
out=$1; shift
args=$1; shift
IFS=';' read -ra CFG <<< "$*"
for c in "${CFG[@]}"; do
  line=$(env $c timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stages']; print('%8.1f Msamples/s  %8.2f ms/step  chunks %d retries %d syncs %d walk_steps %d' % (d['value'], d['ms_per_step'], s['chunks'], s['retries'], s['host_syncs'], s['walk_steps']))
")
  echo "[$c ] $line" | tee -a $out
done
