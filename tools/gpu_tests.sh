set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "icp" -v > gpurun_out/pytest_icp.log 2>&1; rc=$?; echo "pytest icp rc=$rc"; tail -15 gpurun_out/pytest_icp.log; [ $rc -le 1 ] ) &&
( timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -x -rs > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_gpu.log; [ $rc -le 1 ] ) &&
bash tools/gpu_quick.sh
