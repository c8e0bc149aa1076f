set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) && \
( timeout -k 10 300 python tools/run_iss.py 32 > gpurun_out/run_iss.log 2>&1; echo "run_iss rc=$?"; cat gpurun_out/run_iss.log )
