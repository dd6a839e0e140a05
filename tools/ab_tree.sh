#!/bin/bash
# alternate HEAD tree and _ab_old tree, worker-form bench and driver-form bench
for i in 1 2; do
  for d in . _ab_old; do
    ( cd $GRAFT_REPO_ROOT/$d; echo -n "$d worker: "; timeout 300 python bench.py --worker --no-fp32-leg --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['conv_ms_per_step'])" )
  done
done
for d in . _ab_old; do
  ( cd $GRAFT_REPO_ROOT/$d; echo -n "$d driver: "; timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['step_ms'], d['roofline']['avg_launch_ms'])" )
done
