export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r01
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_b256_under_rocprof.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/bench_b256_kernel_stats.csv
run_pmc() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -- python3 bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > $O/pmc_$n.log 2>&1
}
run_pmc sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_b64_sq.txt
python tools/pmc_clock.py $O/pmc_sq f16x3 > $O/pmc_b64_clock_f16x3.txt
run_pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
python tools/pmc_summary.py $O/pmc_lds > $O/pmc_b64_lds.txt
run_pmc fetch FETCH_SIZE
python tools/pmc_summary.py $O/pmc_fetch > $O/pmc_b64_fetch.txt
run_pmc write WRITE_SIZE
python tools/pmc_summary.py $O/pmc_write > $O/pmc_b64_write.txt
rm -rf $O/stats $O/pmc_sq $O/pmc_lds $O/pmc_fetch $O/pmc_write $O/*.log $O/stats.err
