set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/soak_nn1.py 15000 31000000 > gpurun_out/soak.txt 2>&1; rc=$?; tail -1 gpurun_out/soak.txt; [ $rc -eq 0 ] || exit 1
PCR_SWEEP_SCALE=25 PCR_SWEEP_SEED=777001 timeout -k 10 700 python -m pytest tests/test_random_sweeps.py -m gpu -q -x --timeout 650 > gpurun_out/sweep.txt 2>&1; tail -2 gpurun_out/sweep.txt
