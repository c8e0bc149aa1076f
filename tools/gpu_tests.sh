set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_random_sweeps.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) || exit 1
for rep in 1 2 3; do for lib in a hip; do ( PCR_LIB_PATH=$L/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'M corr/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4))" ) || exit 1; done; done
for v in 1 2 3; do ( PCR_LIB_PATH=$L/libpcr_a.so NN_METHOD=1 timeout -k 10 200 python tools/run_nn1.py 120000 10 $v 2>&1 | tail -1 | sed 's/^/old /'; NN_METHOD=1 timeout -k 10 200 python tools/run_nn1.py 120000 10 $v 2>&1 | tail -1 | sed 's/^/new /' ) || exit 1; done
