set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_iss.py -m gpu -q -x -rs > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) && \
( timeout -k 10 300 python tools/run_iss.py 4 8 16 32 > gpurun_out/run_iss.log 2>&1; echo "run_iss rc=$?"; cat gpurun_out/run_iss.log )
