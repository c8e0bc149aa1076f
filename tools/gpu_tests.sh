set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd
for n in 120000 2000000; do for lib in a hip a hip; do ( echo "== $lib n=$n"; PCR_LIB_PATH=$L/libpcr_$lib.so timeout -k 10 300 python tools/run_knn.py $n 2>&1 | grep -E "^grid k-NN" | sed 's/ (cell scale auto), [0-9]* x [0-9]*: kernel/:/; s/ -> .*//' | tr '\n' ' '; echo ) || exit 1; done; done
