"""N > 1 path on CPU: two gloo ranks, each an independent replica (own prompts, own seed), no data-path collective;
only the scalar timing aggregation goes through the process group."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from longlive_amd import replicas, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prompts = [f"prompt {i}" for i in range(7)]
    mine = replicas.shard_prompts(prompts, rank, world)
    cfg = synth.toy_config()
    noise = synth.synth_noise(cfg, 2, seed=replicas.replica_seed(0, rank))
    frames, elapsed = 12.0 * (rank + 1), 0.5 + 0.25 * rank
    dist.barrier()
    tot, tmax = replicas.aggregate_throughput(frames, elapsed)
    out.put((rank, mine, float(noise.float().sum()), tot, tmax))
    dist.destroy_process_group()


def test_two_replicas_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, n0, tot0, t0), (r1, m1, n1, tot1, t1) = res
    assert m0 == ["prompt 0", "prompt 2", "prompt 4"] and m1 == ["prompt 1", "prompt 3", "prompt 5"]   # drop_last
    assert n0 != n1                          # different seeds -> different streams
    assert tot0 == tot1 == 36.0 and t0 == t1 == 0.75      # whole-job frames / max-over-ranks time


def test_single_replica_needs_no_process_group():
    assert replicas.aggregate_throughput(12.0, 0.5) == (12.0, 0.5)
    assert replicas.shard_prompts(list(range(5)), 0, 1) == [0, 1, 2, 3, 4]
