#!/bin/bash
# on the GPU box: alternate A (libmodex_A.so) and B (the current build) N times on one config; prints the effect kernels' isolated launch times and floor fractions
# usage: tools/ab_fx.sh <config> [N]
C=${1:-5}; N=${2:-2}
for i in $(seq $N); do
  for v in A B; do
    if [ $v = A ]; then export MODEX_HIP_LIB=$PWD/mod_extraction_amd/_lib/libmodex_A.so; else unset MODEX_HIP_LIB; fi
    echo -n "$v "; timeout 300 python bench.py --worker --config $C --no-fp32-leg --steps 8 --warmup 2 --no-cpu-baseline --detail-out /tmp/ab_detail.json >/dev/null 2>&1; python -c "
import json
d=json.load(open('/tmp/ab_detail.json')); k=d.get('fx_kernels') or d['kernels']
print(round(d['ms_per_step'],3), {n: (v['avg_launch_ms'], v.get('frac_of_independent_floor'), v.get('frac_of_serial_floor')) for n, v in k.items() if n.endswith('_kernel') and 'avg_launch_ms' in v})"
  done
done
