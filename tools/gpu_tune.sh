set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_gpu.log; [ $rc -le 1 ] ) &&
( timeout -k 10 300 python tools/tune_nn1.py 120000 5 > gpurun_out/tune2.txt 2>&1; echo "tune rc=$?"; head -30 gpurun_out/tune2.txt )
