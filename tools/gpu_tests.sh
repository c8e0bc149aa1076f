set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_random_sweeps.py tests/test_multirank.py tests/test_config1.py -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 600 python tools/soak_nn1.py 1500 777000 | tail -2 || exit 1
for lib in old hip old hip; do ( PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline --no-grid-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4))" ) || exit 1; done
for lib in old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 200 python tools/run_hw9.py 120000 1 1 | sed "s/^/$lib /" || exit 1; done
