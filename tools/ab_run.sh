#!/bin/bash
# on the GPU box: alternate A (libmodex_A.so) and B (the current build) N times; prints audio-s/s, ms/step, conv ms and the kernels named in $KEYS
# (bench.py's stdout carries only the short headline line: the per-kernel table is read from the --detail-out file)
N=${1:-2}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = A ]; then export MODEX_HIP_LIB=$PWD/mod_extraction_amd/_lib/libmodex_A.so; else unset MODEX_HIP_LIB; fi
    echo -n "$v "; timeout 300 python bench.py --worker --no-fp32-leg --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --detail-out /tmp/ab_detail.json ${BENCH_ARGS} >/dev/null 2>&1; python -c "
import json,sys,os
d=json.load(open('/tmp/ab_detail.json')); print(round(d['value']), round(d['ms_per_step'],2), d.get('conv_ms_per_step'), {k: d['kernels'][k]['avg_ms'] for k in os.environ.get('KEYS','').split() if k in d['kernels']})"
  done
done
