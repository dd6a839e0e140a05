#!/bin/bash
# on the GPU box: average duration of the kernels matching $1 in a 4-step headline run
export TMPDIR=/tmp
rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -- python3 bench.py --worker --config ${2:-3} --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg > /tmp/kst.json 2>/tmp/kst.err
python3 - "$1" <<'PY'
import csv, glob, json, re, sys
f = glob.glob('/tmp/kst/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[1], r['Name']):
        print(f"{r['Name'][:70]:70s} x{r['Calls']:>4s}  {float(r['AverageNs'])/1e3:9.1f} us")
d = json.loads(open('/tmp/kst.json').read().strip().splitlines()[-1]); print('step ms', round(d['ms_per_step'], 2))
PY
