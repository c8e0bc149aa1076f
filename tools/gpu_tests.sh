set -o pipefail
timeout -k 10 300 python tools/gpu_scratch.py
