set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd
for rep in 1 2; do for lib in a hip; do
( export PCR_LIB_PATH=$L/libpcr_$lib.so
  a=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 120000 40 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  a1=$(NN_METHOD=2 ALIGNED=0 timeout -k 10 200 python tools/run_nn1.py 120000 40 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  a2=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 30000 40 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  d=$(timeout -k 10 300 python bench.py --no-cpu-baseline --nn grid 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid-icp-120k ms/step', round(d['ms_per_step'],4), 'avg nn', round(d['roofline']['avg_launch_ms_over_the_timed_icp'],4))")
  echo "$lib | 120k aligned $a misaligned $a1 | 30k $a2 | $d" ) || exit 1
done; done
