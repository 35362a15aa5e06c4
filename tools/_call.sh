set -u
mkdir -p gpurun_out
FENNEC_BENCH_BACKEND=gloo FENNEC_BENCH_SINGLE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --batch 8 --batch-items 96 --batch-files 8 --no-extras --no-cpu-baseline > gpurun_out/c17_two_rank.json 2> gpurun_out/c17_two_rank.err; echo "rc $?"
tail -c 600 gpurun_out/c17_two_rank.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/c17_two_rank.json').read().strip().splitlines()[-1])
print(d['n_gpus'], d['value'], d['scaling'])
print(json.dumps(d['dist'])[:900])
print(d['batch']['value'], d['batch']['summarize'])
PY
