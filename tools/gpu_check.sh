# Round check on the GPU box: smoke, GPU parity tests, bench, rocprofv3 kernel stats + PMC passes.
# usage (from the build container):  gpurun --timeout 1200 -- 'bash tools/gpu_check.sh'
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; echo "smoke rc=$rc"; tail -2 gpurun_out/smoke.log; [ $rc -eq 0 ] ) &&
( timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/pytest_gpu.log; [ $rc -le 1 ] ) &&
( timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?; echo "bench rc=$rc"; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err; [ $rc -eq 0 ] ) &&
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1; echo "rocprof stats rc=$?" ) &&
export ICP_LOOP=1
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_sq --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/pmc_sq.log 2>&1; echo "pmc sq rc=$?" ) &&
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/pmc_sq2 --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/pmc_sq2.log 2>&1; echo "pmc sq2 rc=$?" ) &&
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?" ) &&
( cd /tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?" )
