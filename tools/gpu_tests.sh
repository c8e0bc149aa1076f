set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
export ICP_LOOP=1
cd /tmp
( timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/pmc_m2 --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/pmc_m2.log 2>&1; echo "pmc2 rc=$?"; grep -i "unable\|missing" $R/gpurun_out/pmc_m2.log | cut -c1-400 )
