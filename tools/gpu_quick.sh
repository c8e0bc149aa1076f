set -o pipefail
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --nn grid --steps 20 2>&1 | tee gpurun_out/bench_grid.json | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print({k: d[k] for k in ('value','icp_iter_per_s','ms_per_step')}, d['roofline']['avg_launch_ms'], d['config']['pose_err_vs_gt_fro'])
    except Exception as e: print(l[:300])
"
python bench.py --no-cpu-baseline --nn grid --steps 40 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print({k: d[k] for k in ('value','icp_iter_per_s','ms_per_step')}, d['roofline']['avg_launch_ms'], d['config']['pose_err_vs_gt_fro'])
    except Exception as e: print(l[:300])
"
