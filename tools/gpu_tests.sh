set -o pipefail
timeout -k 10 300 python tools/run_real_scan.py 20
