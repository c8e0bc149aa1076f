set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python tools/tune_nn1.py 120000 5 2>&1 | tee gpurun_out/tune3.txt | head -20 )
