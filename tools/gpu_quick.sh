set -o pipefail
mkdir -p gpurun_out
python bench.py --no-cpu-baseline 2>gpurun_out/b1.err | tee gpurun_out/b1.json | cut -c1-400
python bench.py --no-cpu-baseline --nn grid 2>gpurun_out/b2.err | tee gpurun_out/b2.json | cut -c1-1800
python bench.py --no-cpu-baseline --workload c5 --steps 10 --warmup 1 2>gpurun_out/b3.err | tee gpurun_out/b3.json | cut -c1-1800
export HSA_ENABLE_IPC_MODE_LEGACY=0
PCR_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline --workload c5 --points 400000 2> gpurun_out/b4.err | tee gpurun_out/b4.json | cut -c1-700
tail -2 gpurun_out/b1.err gpurun_out/b2.err gpurun_out/b3.err gpurun_out/b4.err | grep -v amdgpu.ids
