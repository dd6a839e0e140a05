#!/bin/bash
# usage: ab_env.sh VAR N  -> alternate VAR=0 / VAR=1
VAR=$1; N=${2:-2}
for i in $(seq $N); do
  for v in 0 1; do
    export $VAR=$v
    echo -n "$VAR=$v "; timeout 300 python bench.py --worker --no-fp32-leg --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['conv_ms_per_step'])"
  done
done
