set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "bounded or icp" 2>&1 | tail -3
