set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/soak_nn1.py 15000 8800000 > gpurun_out/soak.txt 2>&1; rc=$?; tail -2 gpurun_out/soak.txt; grep -c MISMATCH gpurun_out/soak.txt; exit $rc
