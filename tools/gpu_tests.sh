set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python tools/run_c5.py 10000000 20 > gpurun_out/run_c5.log 2>&1; echo "rc=$?"; cat gpurun_out/run_c5.log ) && \
( timeout -k 10 300 python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c5.json 2> gpurun_out/bench.err; echo "bench c5 rc=$?"; cat gpurun_out/bench_c5.json )
