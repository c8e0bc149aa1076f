set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_knn_grid.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] )
