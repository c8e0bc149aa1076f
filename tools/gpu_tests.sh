set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
for lib in old hip old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 200 python tools/run_hw9.py 4000 800 0 | sed "s/^/$lib /" || exit 1; done
for lib in old hip old hip; do ( PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python bench.py --no-cpu-baseline --nn grid 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib 120k grid', 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'pose err', d['config']['pose_err_vs_gt_fro'])" ) || exit 1; done
timeout -k 10 600 python tools/soak_nn1.py 600 90000 | tail -2
