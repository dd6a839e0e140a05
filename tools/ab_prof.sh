#!/bin/bash
# on the GPU box: per-kernel time of A (libmodex_A.so) and B (the current build) from rocprofv3 --kernel-trace --stats
# usage: tools/ab_prof.sh '<grep -E pattern on kernel names>' [batch]
export TMPDIR=/tmp
PAT=${1:-.}; BATCH=${2:-256}
for v in A B; do
  if [ $v = A ]; then export MODEX_HIP_LIB=$PWD/mod_extraction_amd/_lib/libmodex_A.so; else unset MODEX_HIP_LIB; fi
  rm -rf gpurun_out/abp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$v -- python3 bench.py --worker --steps 2 --warmup 1 --batch $BATCH --no-cpu-baseline > gpurun_out/abp_$v.log 2>&1
  f=$(find gpurun_out/abp_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - "$f" "$PAT" <<'PY'
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if re.search(sys.argv[2], r["Name"])]
for r in rows:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>4s} total_ms={int(r['TotalDurationNs'])/1e6:9.3f} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
done
