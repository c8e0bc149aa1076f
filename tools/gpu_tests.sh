set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_voxel_filter.py tests/test_random_sweeps.py -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
for lib in old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python tools/run_n3.py 2>&1 | grep "voxel_filter" | cut -c1-150 | sed "s/^/$lib /" || exit 1; done
