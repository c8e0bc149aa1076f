set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "icp or grid" > gpurun_out/pytest_icp.log 2>&1; rc=$?; echo "pytest icp/grid rc=$rc"; tail -5 gpurun_out/pytest_icp.log; [ $rc -le 1 ] ) &&
( python bench.py --no-cpu-baseline 2>gpurun_out/bench.err | tee gpurun_out/bench_quick.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','icp_iter_per_s','ms_per_step')}, d['roofline']['frac'], d['exact_grid'])" ) &&
( timeout -k 10 600 python tools/run_c5.py 10000000 20 2>&1 | tail -3 | tee gpurun_out/c5_10m_warm.txt )
