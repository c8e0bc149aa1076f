set -o pipefail
PCR_SWEEP_SCALE=40 PCR_SWEEP_SEED=123456 timeout -k 10 1000 python -m pytest tests/test_random_sweeps.py -m gpu -q -x --timeout 900 2>&1 | tail -5 &&
timeout -k 10 300 python -m pytest tests/test_random_sweeps.py -m gpu -q -x 2>&1 | tail -3
