set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_random_sweeps.py tests/test_multirank.py -m gpu -q -x --timeout 500 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for lib in old hip; do ( PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-grid-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4))" ) || exit 1; done; done
export TMPDIR=/tmp ICP_LOOP=1
cd /tmp
( timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?" ) &&
( timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write --output-format csv -- python3 $R/tools/run_nn1.py 120000 9 > $R/gpurun_out/pmc_write.log 2>&1; echo "pmc write rc=$?" )
