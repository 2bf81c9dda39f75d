#!/usr/bin/env python3
"""Throughput of N independent B = 1 prompt streams interleaved on N HIP streams of one process (pipeline/throughput.py), N = 1 .. 4,
on bench.py's steady-state workload: how far the side-by-side gain of the two-stream mode carries.  Interleaved rounds (every N
measured in every round) so that device drift does not pass for a difference.

    python3 tools/nstreams.py [--rounds 3] [--blocks 6] [--quant int8] [--max 4]   ->  one JSON line
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import bench
from longlive_amd import _lib, synth
from longlive_amd.pipeline import CausalInferencePipeline, InterleavedStreams
from longlive_amd.wan_wrapper import WanDiffusionWrapper


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--max", type=int, default=4)
    ap.add_argument("--quant", default="none")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _lib.load()
    cfg = synth.longlive_1_3b(local_attn_size=12, sink_size=3)
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=12, sink_size=3, cfg=cfg, device=dev,
                              state_dict=synth.synth_state_dict(cfg, seed=0, device=dev))
    gen.model.set_quant(None if a.quant == "none" else a.quant)
    res = {n: [] for n in range(1, a.max + 1)}
    with torch.no_grad():
        for r in range(a.rounds):
            for n in range(1, a.max + 1):
                pipes = [CausalInferencePipeline(bench._pipe_args(), dev, generator=gen) for _ in range(n)]
                runner = InterleavedStreams(pipes, dev)
                noises = [synth.synth_noise(cfg, 3 * (4 + a.blocks), seed=s, device=dev) for s in range(n)]
                prompts = [{"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1 + s, device=dev)} for s in range(n)]
                st = runner.stream(noises, prompts)
                for _ in range(4):
                    next(st)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.blocks):
                    next(st)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                res[n].append(n * 12 * a.blocks / dt)
                del st, runner, pipes, noises
                torch.cuda.empty_cache()
                print(f"round {r} streams {n}: {res[n][-1]:.2f} frames/s", file=sys.stderr, flush=True)
    out = {"workload": "bench.py steady-state blocks, N independent B = 1 streams on N HIP streams", "quant": a.quant, "rounds": a.rounds,
           "blocks": a.blocks, "fps": {str(n): sorted(v)[len(v) // 2] for n, v in res.items()}, "all": {str(n): v for n, v in res.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
