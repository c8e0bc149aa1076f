set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python -m pytest tests/test_global_registration.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) && \
( timeout -k 10 300 python tools/run_n4.py > gpurun_out/run_n4.log 2>&1; echo "run_n4 rc=$?"; cat gpurun_out/run_n4.log )
