set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd
( timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) || exit 1
for rep in 1 2; do for lib in a hip; do
( export PCR_LIB_PATH=$L/libpcr_$lib.so
  a=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 120000 40 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  a2=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 1000000 10 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  b=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 10000000 3 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  c=$(timeout -k 10 300 python tools/run_c5.py 10000000 20 2>&1 | grep -o "ICP 20 iterations: [0-9.]* ms total, [0-9.]* ms/iter")
  d=$(timeout -k 10 300 python bench.py --no-cpu-baseline --nn grid 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid-icp-120k ms/step', round(d['ms_per_step'],4), 'avg nn', round(d['roofline']['avg_launch_ms_over_the_timed_icp'],4))")
  echo "$lib | 120k $a | 1M $a2 | 10M $b | $c | $d" ) || exit 1
done; done
