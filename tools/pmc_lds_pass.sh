# on the GPU box (through gpurun): LDS-side and VALU-side counters of the headline step's kernels at bs 64
#   bash tools/pmc_lds_pass.sh r05       -> gpurun_out/r05/pmc_b64_lds.txt, pmc_b64_valu.txt
# one counter group per pass, --kernel-trace only (no other trace domain beside --pmc).
set -u
: "${1:?usage: tools/pmc_lds_pass.sh <rNN>}"
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/$1
mkdir -p $O
run_pmc() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -- python3 bench.py --worker --steps 2 --warmup 1 --batch 64 --no-cpu-baseline --no-fp32-leg > $O/pmc_$n.log 2>&1
}
run_pmc lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
python tools/pmc_summary.py $O/pmc_lds > $O/pmc_b64_lds.txt
run_pmc valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
python tools/pmc_summary.py $O/pmc_valu > $O/pmc_b64_valu.txt
rm -rf $O/pmc_lds $O/pmc_valu
tail -3 $O/pmc_lds.log $O/pmc_valu.log
ls -la $O
