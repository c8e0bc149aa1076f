set -o pipefail
mkdir -p gpurun_out
export PCR_BENCH_BACKEND=gloo
( timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 5 --warmup 1 > gpurun_out/bench_r4.json 2> gpurun_out/bench_r4.err; echo "rehearsal c2 x4 rc=$?"; tail -c 1500 gpurun_out/bench_r4.json; echo; tail -3 gpurun_out/bench_r4.err ) && \
( timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 4 --steps 5 --warmup 1 --workload c5 --points 2000000 > gpurun_out/bench_r4c5.json 2> gpurun_out/bench_r4c5.err; echo "rehearsal c5 x4 rc=$?"; head -c 900 gpurun_out/bench_r4c5.json; echo; tail -3 gpurun_out/bench_r4c5.err )
