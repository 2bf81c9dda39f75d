# rehearsal of both N > 1 launch paths on the one-GPU box: 2 replicas sharing the card (each runs at about half speed)
LL_BENCH_SHARE_GPUS=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 4 --no-extras > gpurun_out/launch2.json 2> gpurun_out/launch2.err; echo "launcher rc=$?"; tail -3 gpurun_out/launch2.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/launch2.json")); print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, d["config"]["per_replica_fps"], d["config"]["parallelism"], d["roofline"]["avg_us"] if d["roofline"] else None)
PY
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/torchrun1.json 2> gpurun_out/torchrun1.err; echo "torchrun rc=$?"; tail -2 gpurun_out/torchrun1.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/torchrun1.json")); print({k:d[k] for k in ("value","n_gpus","ms_per_step")}, d["config"]["per_replica_fps"])
PY
