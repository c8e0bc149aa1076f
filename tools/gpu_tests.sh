set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
