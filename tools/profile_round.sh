# rocprofv3 summaries of one round (run on the GPU box through gpurun):  bash tools/profile_round.sh r02
# kernel statistics of every BASELINE config (bench.py --config 2..5) + the PMC passes of the headline roofline kernel.
set -eu
: "${1:?usage: tools/profile_round.sh <rNN>}"
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
R=$1
O=gpurun_out/$R
rm -rf $O; mkdir -p $O
ONLY=${2:-all}
stats() { c=$1; steps=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c$c -- python3 bench.py --worker --config $c --steps $steps --warmup 1 --no-cpu-baseline --no-fp32-leg --detail-out $O/bench_c${c}_under_rocprof_detail.json > $O/bench_c${c}_under_rocprof.json 2> $O/stats_c$c.err \
    || echo "config $c: rocprofv3 exited with $? (after writing its CSVs; seen at process teardown with CU-masked streams)" >> $O/notes.txt
  cp $(find $O/stats_c$c -name "*kernel_stats.csv" | head -1) $O/bench_c${c}_kernel_stats.csv
  rm -rf $O/stats_c$c
}
if [ $ONLY = all ] || [ $ONLY = stats ]; then stats 3 5; stats 2 10; stats 4 2; stats 5 5; fi
if [ $ONLY = stats ]; then rm -f $O/*.err; ls -la $O; exit 0; fi
run_pmc() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -- python3 bench.py --worker --steps 2 --warmup 1 --batch 64 --no-cpu-baseline --no-fp32-leg > $O/pmc_$n.log 2>&1
}
run_pmc sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
python tools/pmc_summary.py $O/pmc_sq > $O/pmc_b64_sq.txt
python tools/pmc_clock.py $O/pmc_sq f16x3 > $O/pmc_b64_clock_f16x3.txt
run_pmc fetch FETCH_SIZE
python tools/pmc_summary.py $O/pmc_fetch > $O/pmc_b64_fetch.txt
run_pmc write WRITE_SIZE
python tools/pmc_summary.py $O/pmc_write > $O/pmc_b64_write.txt
# the MR-STFT entry point of config 5 at bs 64 (FETCH_SIZE / WRITE_SIZE in separate passes) -> pmc_mrstft.json for bench.py's roofline.traffic
run_pmc5() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc5_$n -- python3 bench.py --worker --config 5 --steps 2 --warmup 1 --batch 64 --no-cpu-baseline > $O/pmc5_$n.log 2>&1
}
run_pmc5 fetch FETCH_SIZE
python tools/pmc_summary.py $O/pmc5_fetch mr_ > $O/pmc_c5_b64_fetch.txt
run_pmc5 write WRITE_SIZE
python tools/pmc_summary.py $O/pmc5_write mr_ > $O/pmc_c5_b64_write.txt
run_pmc5 valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
python tools/pmc_summary.py $O/pmc5_valu mr_ > $O/pmc_c5_b64_valu.txt
python tools/pmc_traffic_json.py $O
rm -rf $O/pmc5_fetch $O/pmc5_write $O/pmc5_valu
# the sample-recurrent kernels of config 4 (LSTM forward / backward): issue and LDS counters
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lstm -- python3 tools/bench_lstm.py > $O/pmc_lstm.log 2>&1
python tools/pmc_summary.py $O/pmc_lstm lstm > $O/pmc_lstm_sq.txt
# LDS-side and VALU-side counters of the headline step's kernels (the limiter table of profiles/rNN/README.md)
run_pmc lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
python tools/pmc_summary.py $O/pmc_lds > $O/pmc_b64_lds.txt
run_pmc valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
python tools/pmc_summary.py $O/pmc_valu > $O/pmc_b64_valu.txt
rm -rf $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_lstm $O/pmc_lds $O/pmc_valu $O/*.log $O/*.err
ls -la $O
