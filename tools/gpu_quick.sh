set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
PCR_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline 2> gpurun_out/rehearse.err | tee gpurun_out/rehearse2.json | cut -c1-900
tail -3 gpurun_out/rehearse.err
