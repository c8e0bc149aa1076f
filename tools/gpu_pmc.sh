set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for order in scan shuffle; do for v in 0 1; do python3 $R/tools/run_nn1.py 120000 5 $v 2 8 $order; done; done > $R/gpurun_out/order.txt 2>&1
cat $R/gpurun_out/order.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_a --output-format csv -- python3 $R/tools/run_nn1.py 120000 3 0 2 8 > $R/gpurun_out/pmc_a.log 2>&1; echo "pmc_a rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace -d $R/gpurun_out/pmc_b --output-format csv -- python3 $R/tools/run_nn1.py 120000 3 0 2 8 > $R/gpurun_out/pmc_b.log 2>&1; echo "pmc_b rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_c --output-format csv -- python3 $R/tools/run_nn1.py 120000 3 1 2 8 > $R/gpurun_out/pmc_c.log 2>&1; echo "pmc_c rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch --output-format csv -- python3 $R/tools/run_nn1.py 120000 3 0 2 8 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc_fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write --output-format csv -- python3 $R/tools/run_nn1.py 120000 3 0 2 8 > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc_write rc=$?"
ls $R/gpurun_out/pmc_a/*/ | head
