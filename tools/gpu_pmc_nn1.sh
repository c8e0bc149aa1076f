# PMC passes over the headline search kernel inside a 9-iteration ICP loop (120 000 x 120 000): the counter part of tools/gpu_check.sh alone.
# usage: gpurun --timeout 900 -- 'bash tools/gpu_pmc_nn1.sh'   then   python tools/summarize_prof.py rNN
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
export ICP_LOOP=1
cd /tmp || exit 1
run() { local d=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/$d --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/$d.log 2>&1; local rc=$?; echo "$d rc=$rc"; tail -1 $R/gpurun_out/$d.log; return $rc; }
run pmc_sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1
run pmc_sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
run pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA || exit 1
run pmc_sq3 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
