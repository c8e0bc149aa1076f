set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_random_sweeps.py -m gpu -q -x --timeout 500 2>&1 | tail -3 || exit 1
for rep in 1 2; do
for lib in old hip; do ( PCR_LIB_PATH=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-grid-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4))" ) || exit 1; done; done
timeout -k 10 300 ./tools/ubench/valu_rate > gpurun_out/valu_rate.txt; grep "W=4" gpurun_out/valu_rate.txt | cut -c1-100
