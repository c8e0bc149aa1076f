set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 600 python tools/soak_nn1.py 3000 4440000 > gpurun_out/soak.txt 2>&1; tail -1 gpurun_out/soak.txt
for lib in old hip; do PCR_LIB_PATH=$R/hands-on-point-cloud-processing_amd/libpcr_$lib.so timeout -k 10 300 python tools/run_outliers.py 120000 0.1 10 | sed "s/^/$lib /" || exit 1; done
