# HBM-side traffic of the configs[3] radius kernels (separate --pmc passes, no other tracing) -> profiles/latest_pmc_c4.json via summarize_c4_pmc.py
# usage: gpurun --timeout 900 -- 'bash tools/gpu_pmc_c4.sh && python tools/summarize_c4_pmc.py r04'
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
pass() { local d=$1; shift
    ( cd /tmp && timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/$d --output-format csv -- python3 $R/tools/run_c4_radius.py > $R/gpurun_out/$d.log 2>&1; rc=$?; echo "$d rc=$rc"; tail -1 $R/gpurun_out/$d.log; exit $rc ) }
pass pmc_c4_fetch FETCH_SIZE && pass pmc_c4_write WRITE_SIZE
