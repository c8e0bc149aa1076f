set -o pipefail
for a in "--steps 1 --warmup 0" "--steps 3 --warmup 1 --no-grid-extra" "--steps 2 --warmup 0 --nn grid" "--steps 20 --warmup 3"; do
  timeout -k 10 300 python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a ->', round(d['value'],1), d['steps'], d['warmup'], round(d['ms_per_step'],3))" || { echo "FAILED: $a"; exit 1; }
done
PCR_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 3 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks gloo rehearsal ->', round(d['value'],1), d['n_gpus'])"
