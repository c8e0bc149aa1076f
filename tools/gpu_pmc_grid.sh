# HBM-side traffic and instruction counts of the exact grid search at the converged pose, inside an ICP loop (separate --pmc passes,
# no other tracing).  usage: gpurun --timeout 1100 -- 'bash tools/gpu_pmc_grid.sh [n=10000000] [suffix] [PCR_TUNE string]'
#   e.g.  bash tools/gpu_pmc_grid.sh 10000000 _walk grid_tile=2      (the cell walk alone, for the before / after table)
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
N=${1:-10000000}
SUF=${2:-}
# the kernels as the ICP loop runs them: sorted working cloud, gate bound, record-position seeds (search 0 = the one-shot search that
# builds the index, search 1 = the cold first search of the loop: the summary skips both)
export TMPDIR=/tmp NN_METHOD=2 ALIGNED=1 ICP_LOOP=1 PCR_TUNE=${3:-}
pass() {   # pass <dir> <counters...>
    local d=$1; shift
    ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/$d$SUF --output-format csv -- python3 $R/tools/run_nn1.py $N 6 > $R/gpurun_out/$d$SUF.log 2>&1; rc=$?; echo "$d$SUF rc=$rc"; tail -1 $R/gpurun_out/$d$SUF.log; exit $rc )
}
pass pmc_grid_fetch FETCH_SIZE &&
pass pmc_grid_write WRITE_SIZE &&
pass pmc_grid_sq SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE
