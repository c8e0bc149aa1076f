set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 3 2; do
NN_METHOD=$m ALIGNED=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace -d $R/gpurun_out/pmc_m$m --output-format csv -- python3 $R/tools/run_nn1.py 120000 3 > $R/gpurun_out/pmc_m$m.log 2>&1; echo "pmc m$m rc=$?"
NN_METHOD=$m ALIGNED=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --kernel-trace -d $R/gpurun_out/pmc2_m$m --output-format csv -- python3 $R/tools/run_nn1.py 120000 3 > $R/gpurun_out/pmc2_m$m.log 2>&1; echo "pmc2 m$m rc=$?"
done
