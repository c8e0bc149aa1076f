set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3 || exit 1
timeout -k 10 600 python tools/soak_nn1.py 3000 1230000 > gpurun_out/soak.txt 2>&1; tail -1 gpurun_out/soak.txt
timeout -k 10 200 python tools/run_hw9.py 120000 1 1
timeout -k 10 200 python tools/run_hw9.py 120000 20 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-grid-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],2), round(d['ms_per_step'],4))"
