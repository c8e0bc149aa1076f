set -o pipefail
mkdir -p gpurun_out
for nn in grid brute; do timeout -k 10 300 python bench.py --no-cpu-baseline --points 4000 --nn $nn --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$nn n=4000: ms/step', round(d['ms_per_step'],4), 'iter/s', round(d['icp_iter_per_s']), 'nn kernel ms', round(d['roofline'].get('avg_launch_ms_over_the_timed_icp', d['roofline'].get('avg_launch_ms')),4))" || exit 1; done
