set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python -m pytest tests/test_voxel_filter.py -m gpu -q -x > gpurun_out/pytest_vox.log 2>&1; rc=$?; echo "pytest voxel rc=$rc"; tail -15 gpurun_out/pytest_vox.log; [ $rc -le 1 ] )
