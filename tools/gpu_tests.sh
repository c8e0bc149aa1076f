set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/hands-on-point-cloud-processing_amd
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_config5.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/pytest_gpu.log; [ $rc -eq 0 ] ) || exit 1
for rep in 1 2; do for lib in a hip; do
( export PCR_LIB_PATH=$L/libpcr_$lib.so
  a=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 120000 40 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  a2=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 1000000 10 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  a3=$(NN_METHOD=2 ALIGNED=0 timeout -k 10 200 python tools/run_nn1.py 1000000 10 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  b=$(NN_METHOD=2 ALIGNED=1 timeout -k 10 200 python tools/run_nn1.py 10000000 3 2>&1 | tail -1 | grep -o "nn1_grid: [0-9.]* ms")
  c=$(timeout -k 10 300 python tools/run_c5.py 10000000 20 2>&1 | grep -o "ICP 20 iterations: [0-9.]* ms total, [0-9.]* ms/iter")
  echo "$lib | 120k $a | 1M aligned $a2 misaligned $a3 | 10M $b | $c" ) || exit 1
done; done
