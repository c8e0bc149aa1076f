# HBM-side traffic of the exact grid search at the converged pose, inside an ICP loop (separate --pmc passes, no other tracing):
# usage: gpurun --timeout 900 -- 'bash tools/gpu_pmc_grid.sh [n=10000000]'
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
N=${1:-10000000}
# the kernel as the ICP loop runs it: sorted working cloud, gate bound, record-position seeds (launch 0 = the one-shot search that
# builds the index, launch 1 = the cold first search of the loop: the summary skips both)
export TMPDIR=/tmp NN_METHOD=2 ALIGNED=1 ICP_LOOP=1
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_grid_fetch --output-format csv -- python3 $R/tools/run_nn1.py $N 5 > $R/gpurun_out/pmc_grid_fetch.log 2>&1; echo "pmc grid fetch rc=$?"; tail -1 $R/gpurun_out/pmc_grid_fetch.log ) &&
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_grid_write --output-format csv -- python3 $R/tools/run_nn1.py $N 5 > $R/gpurun_out/pmc_grid_write.log 2>&1; echo "pmc grid write rc=$?"; tail -1 $R/gpurun_out/pmc_grid_write.log ) &&
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_grid_sq --output-format csv -- python3 $R/tools/run_nn1.py $N 5 > $R/gpurun_out/pmc_grid_sq.log 2>&1; echo "pmc grid sq rc=$?"; tail -1 $R/gpurun_out/pmc_grid_sq.log )
