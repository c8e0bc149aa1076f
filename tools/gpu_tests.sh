set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_knn_grid.py tests/test_iss.py tests/test_random_sweeps.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
for lib in old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python tools/bench_hw2.py | grep "^GPU" | sed "s/^/$lib /" || exit 1; done
