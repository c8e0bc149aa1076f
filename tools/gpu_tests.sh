set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_random_sweeps.py -m gpu -q -x --timeout 500 2>&1 | tail -3 || exit 1
for lib in old hip old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 200 python tools/run_nn1.py 120000 10 | grep "n=120000" | sed "s/^/$lib /" || exit 1; done
