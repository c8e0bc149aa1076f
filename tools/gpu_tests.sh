set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 900 python -m pytest tests -m gpu -q -x -rs > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) && \
( timeout -k 10 400 python tools/run_knn.py 120000 > gpurun_out/run_knn.log 2>&1; echo "run_knn rc=$?"; cat gpurun_out/run_knn.log ) && \
( timeout -k 10 300 python tools/bench_hw2.py > gpurun_out/bench_hw2.log 2>&1; echo "bench_hw2 rc=$?"; tail -12 gpurun_out/bench_hw2.log )
