# rocprofv3 kernel statistics of the non-headline configs (tools/bench_paths.py): usage  tools/profile_paths.sh <tag>
set -eu
: "${1:?usage: tools/profile_paths.sh <tag>}"
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/bench_paths.py > $O/bench_paths.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_paths_kernel_stats.csv
rm -rf $O/stats
tail -5 $O/stats.err
cat $O/bench_paths.json
