set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 900 python tools/soak_nn1.py 8000 7770000 > gpurun_out/soak.txt 2>&1; tail -1 gpurun_out/soak.txt; grep -c MISMATCH gpurun_out/soak.txt
PCR_SWEEP_SCALE=10 PCR_SWEEP_SEED=909090 timeout -k 10 600 python -m pytest tests/test_random_sweeps.py -m gpu -q -x 2>&1 | tail -2
timeout -k 10 300 python tools/run_outliers.py 120000 0.1 10
timeout -k 10 300 python tools/run_c5.py 10000000 10 2>&1 | grep "first nn1\|ICP 10\|oracle" | cut -c1-170
