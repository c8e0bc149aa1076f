set -o pipefail
timeout -k 10 900 python tools/soak_etrack.py 3000 50000 | tail -6
