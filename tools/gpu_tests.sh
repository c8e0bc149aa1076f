set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 300 python tools/run_c4.py > gpurun_out/run_c4.log 2>&1; echo "run_c4 rc=$?"; cat gpurun_out/run_c4.log )
