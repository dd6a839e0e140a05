export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 bench.py --worker --config 3 --steps 4 --warmup 1 --no-cpu-baseline --no-fp32-leg > /tmp/tr.json 2>/tmp/tr.err
python3 tools/trace_gaps_c3.py /tmp/tr | tail -24
