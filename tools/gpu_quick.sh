set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/run_c4.py 120000 2>&1 | tee gpurun_out/c4.txt
timeout -k 10 300 python tools/bench_hw2.py 2>&1 | grep -v "points read" | tee gpurun_out/hw2_protocol.txt
( mkdir -p /tmp/hw2run/build && python -c "
import numpy as np
g = np.load('tests/golden/kat_kitti_q5.npz'); rows = np.concatenate([g['db_f32'][:100000], np.zeros((100000,1),np.float32)],axis=1); rows.astype(np.float32).tofile('/tmp/hw2run/000000.bin')" && cd /tmp/hw2run/build && $GRAFT_REPO_ROOT/oracle/_ref/hw2_benchmark_dropin | tail -3 ) 2>&1 | tee -a gpurun_out/hw2_protocol.txt
