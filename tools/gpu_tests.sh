set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for lib in old hip old hip; do ( PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline --nn grid 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib 120k grid', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4))" ) || exit 1; done
for lib in old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 400 python tools/run_c5.py 10000000 10 2>&1 | grep "ICP 10\|first nn1\|oracle" || exit 1; done
