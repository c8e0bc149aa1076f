set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
for lib in old hip old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 200 python tools/run_hw9.py 4000 800 0 | sed "s/^/$lib /" || exit 1; done
for lib in old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python tools/run_n4.py 2>&1 | grep -i "ransac\|hypoth" | head -4 | cut -c1-200 | sed "s/^/$lib /" || exit 1; done
