set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python tools/run_c5.py 10000000 20 > gpurun_out/run_c5.log 2>&1; echo "rc=$?"; cat gpurun_out/run_c5.log ) && \
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_multirank.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] )
