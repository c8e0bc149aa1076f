set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python -m pytest tests/test_multirank.py -m gpu -q -x -rs > gpurun_out/pytest_mr.log 2>&1; rc=$?; echo "pytest mr rc=$rc"; tail -15 gpurun_out/pytest_mr.log; [ $rc -le 1 ] )
