export SIFT_BENCH_BACKEND=gloo MASTER_ADDR=127.0.0.1 SIFT_BENCH_PREFLIGHT=force SIFT_BENCH_PREFLIGHT_TIMEOUT=90
timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29417 bench.py --gpus 2 --steps 2 --warmup 1 --batch 3 --pool 6 --width 320 --height 240 --cap 4096 --no-extras --no-cpu-baseline > gpurun_out/rehearse.json 2> gpurun_out/rehearse.err
echo rc=$?
python3 -c "
import json
l=[x for x in open('gpurun_out/rehearse.json') if x.startswith('{')][-1]
d=json.loads(l); print(json.dumps(d['exchange_preflight'],indent=1)); print(d['exchange'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
