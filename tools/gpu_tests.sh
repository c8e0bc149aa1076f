set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 600 python tools/soak_nn1.py 1500 31337 | tail -1 || exit 1
for lib in old hip old hip; do ( PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'value', round(d['value'],2), 'ms/step', round(d['ms_per_step'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],4), 'grid', round(d['exact_grid']['value'],1), round(d['exact_grid']['ms_per_step'],4))" ) || exit 1; done
