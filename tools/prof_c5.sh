# kernel statistics of BASELINE config 5 (run on the GPU box through gpurun):  bash tools/prof_c5.sh <out dir under gpurun_out>
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-c5}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 bench.py --worker --config 5 --steps 5 --warmup 1 --no-cpu-baseline > $O/c5_prof.json 2> $O/c5_prof.err
cp $(find $O/st -name "*kernel_stats.csv" | head -1) $O/c5_kernel_stats.csv
rm -rf $O/st
