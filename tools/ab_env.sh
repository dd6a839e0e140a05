#!/bin/bash
# on the GPU box: alternate VAR=A / VAR=B N times on the current build; prints audio-s/s, ms/step, conv ms and the kernels named in $KEYS
# usage: tools/ab_env.sh VAR A B [N]     (KEYS="conv_block1_fwd_f16[block1] ..." optional)
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do
  for v in $A $B; do
    export $VAR=$v
    echo -n "$VAR=$v "; timeout 300 python bench.py --worker --no-fp32-leg --steps 8 --warmup 2 --no-cpu-baseline --detail-out /tmp/ab_detail.json >/dev/null 2>&1; python -c "
import json,sys,os
d=json.load(open('/tmp/ab_detail.json')); print(round(d['value']), round(d['ms_per_step'],2), d['conv_ms_per_step'], {k: d['kernels'][k]['avg_ms'] for k in os.environ.get('KEYS','').split() if k in d['kernels']})"
  done
done
