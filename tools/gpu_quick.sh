set -o pipefail
mkdir -p gpurun_out
for n in 120000 1000000; do NN_METHOD=2 GRID_STATS=1 timeout -k 10 120 python tools/run_nn1.py $n 3; NN_METHOD=2 GRID_STATS=1 ALIGNED=1 timeout -k 10 120 python tools/run_nn1.py $n 3; done 2>&1 | grep -v "^first" | tee gpurun_out/grid_stats.txt
