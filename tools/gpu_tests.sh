set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd
( timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) || exit 1
for rep in 1 2 3; do for lib in a hip; do ( PCR_LIB_PATH=$L/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'M corr/s', round(d['value'],2), 'iter/s', round(d['icp_iter_per_s'],1), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4), 'grid', round(d['exact_grid']['value'],1))" ) || exit 1; done; done
