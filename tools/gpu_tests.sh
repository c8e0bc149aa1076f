set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd
for rep in 1 2; do for lib in a hip; do ( echo "== $lib"; PCR_LIB_PATH=$L/libpcr_$lib.so timeout -k 10 300 python tools/run_iss.py 32 2>&1 | grep -E "lanes 32" | sed 's/, [0-9]* keypoints.*//' ) || exit 1; done; done
