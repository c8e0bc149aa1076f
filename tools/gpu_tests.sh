set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] )
