set -o pipefail
mkdir -p gpurun_out
python - <<'PY' || exit 1
import importlib, sys, os, psutil, numpy as np
sys.path.insert(0, os.getcwd())
pcr = importlib.import_module("hands-on-point-cloud-processing_amd")
synth = importlib.import_module("hands-on-point-cloud-processing_amd.synth")
src, tgt = synth.kitti_like_pair(3000)
p = psutil.Process()
for i in range(1501):
    with pcr.Context(0) as ctx:
        cs, ct = ctx.cloud(src), ctx.cloud(tgt)
        ctx.icp_point2point(cs, ct, max_iter=3)
        ctx.nn1(ct, cs)
    if i % 500 == 0: print(i, "rss MB", p.memory_info().rss >> 20, flush=True)
PY
timeout -k 10 1000 python tools/soak_nn1.py 20000 5000000 > gpurun_out/soak.txt 2>&1; rc=$?; tail -3 gpurun_out/soak.txt; exit $rc
