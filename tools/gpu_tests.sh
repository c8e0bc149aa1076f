set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -x -rs > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_gpu.log; [ $rc -le 1 ] ) &&
( timeout -k 10 600 python tools/run_c5.py 10000000 20 2>&1 | tee gpurun_out/c5_10m.txt ) &&
( python bench.py --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_quick.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','icp_iter_per_s','ms_per_step')}, d['roofline']['frac'], d['roofline']['traffic'], d['exact_grid'])" )
