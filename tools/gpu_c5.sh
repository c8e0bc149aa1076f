set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/run_c5.py 1000000 10 2>&1 | tee gpurun_out/c5_1m.txt &&
timeout -k 10 900 python tools/run_c5.py 10000000 10 2>&1 | tee gpurun_out/c5_10m.txt
