set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/bench_hw2.py | tee gpurun_out/hw2.txt
