export TMPDIR=/tmp
for v in 1 0; do
  export MODEX_DIRECT_GRADS=$v
  rm -rf /tmp/st_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$v -- python3 bench.py --worker --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg > /tmp/st_$v.json 2>/tmp/st_$v.err
  f=$(find /tmp/st_$v -name "*kernel_stats.csv" | head -1)
  echo "DIRECT=$v"; grep -E "CUDAFunctor_add|copyBuffer|adamw" $f | cut -c1-60,150-260 | head -5
done
