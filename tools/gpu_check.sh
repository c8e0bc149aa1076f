# Round check on the GPU box: smoke, GPU parity tests, bench, rocprofv3 kernel stats + PMC passes.
# usage (from the build container):  gpurun --timeout 1200 -- 'bash tools/gpu_check.sh'
# Every stage must succeed (rc 0) for the next one to start; the PMC passes are separate rocprofv3 runs (one counter
# group each: FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
export ICP_LOOP=1
stage() {   # stage <name> <timeout> <logfile> <command...>
    local name=$1 limit=$2 log=$3; shift 3
    timeout -k 10 "$limit" "$@" > "$log" 2>&1
    local rc=$?
    echo "$name rc=$rc"; tail -4 "$log"
    return $rc
}
# a line a minute into gpurun_out/ while the stages run (a quiet stage — pytest writing to a block-buffered file — must not read as a hang;
# every stage has its own timeout)
( while true; do date +%T >> gpurun_out/heartbeat.log; sleep 60; done ) &
HB=$!
trap 'kill $HB 2>/dev/null' EXIT
stage smoke 300 gpurun_out/smoke.log python -c "import __graft_entry__ as g; g.smoke()" || exit 1
stage pytest 1000 gpurun_out/pytest_gpu.log python -u -m pytest tests -m gpu -q --timeout 600 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
echo "bench rc=$rc"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err; [ $rc -eq 0 ] || exit 1
cd /tmp || exit 1
stage rocprof-stats 300 $R/gpurun_out/prof_bench.log rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline || exit 1
stage pmc-sq 300 $R/gpurun_out/pmc_sq.log rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_sq --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 || exit 1
stage pmc-sq2 300 $R/gpurun_out/pmc_sq2.log rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/pmc_sq2 --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 || exit 1
stage pmc-mfma 300 $R/gpurun_out/pmc_mfma.log rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA --kernel-trace -d $R/gpurun_out/pmc_mfma --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 || exit 1
stage pmc-fetch 300 $R/gpurun_out/pmc_fetch.log rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 || exit 1
stage pmc-write 300 $R/gpurun_out/pmc_write.log rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 || exit 1
