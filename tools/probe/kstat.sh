#!/bin/bash
# on the GPU box: average duration of the kernels matching $1 in a 4-step headline run
export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 bench.py --worker --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg > /tmp/kst.json 2>/tmp/kst.err
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1)
grep -E "$1" $f | awk -F, '{print $1, $2, $4/1000 " us"}' | cut -c1-120
python3 -c "
import json; d=json.loads(open('/tmp/kst.json').read().strip().splitlines()[-1]); print('step ms', d['ms_per_step'])"
