#!/bin/bash
for v in "" "ACN_LDS_STACK=1 ACN_LDS_MAX=0" ; do
  echo "env: $v"
  env $v timeout -k 10 200 python scripts/time_scene.py diamond 240 135 512 50 2>&1 | grep "iter 1" | cut -c1-60
done
scripts/quick_bench.sh default
ACN_LDS_STACK=0 scripts/quick_bench.sh nostack
