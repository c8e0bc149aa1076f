set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -x -rs > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_gpu.log; [ $rc -le 1 ] )
