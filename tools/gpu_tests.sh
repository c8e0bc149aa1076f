set -o pipefail
NN_METHOD=2 GRID_STATS=1 timeout -k 10 400 python tools/run_nn1.py 10000000 2 | grep "stats\|n=10000000" | cut -c1-200
NN_METHOD=2 GRID_STATS=1 ALIGNED=1 timeout -k 10 400 python tools/run_nn1.py 10000000 2 | grep "stats\|n=10000000" | cut -c1-200
