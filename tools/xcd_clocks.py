#!/usr/bin/env python3
"""Per-XCD graphics clocks while bench.py's steady-state workload runs: amdsmi's gpu_metrics carries `current_gfxclks` (one entry per
XCD); a side thread samples it every 50 ms over a few seconds of the pipeline.  Question: do the eight XCDs of a device hold the same
clock (the self-attention launch ends when its slowest workgroup ends)?

    python3 tools/xcd_clocks.py [--blocks 24]   ->  one JSON line
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import bench
from longlive_amd import _lib, synth
from longlive_amd.pipeline import CausalInferencePipeline
from longlive_amd.wan_wrapper import WanDiffusionWrapper


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=24)
    ap.add_argument("--load", default="pipe", choices=["pipe", "gemm", "attn"],
                    help="pipe: the pipeline's steady state; gemm / attn: ONE kernel looped (FFN1 / self-attention at the production shape) -- "
                         "the same work on every CU, to tell a property of the device from a property of the pipeline's work distribution")
    a = ap.parse_args()
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    _lib.load()
    cfg = synth.longlive_1_3b(local_attn_size=12, sink_size=3)
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=12, sink_size=3, cfg=cfg, device=dev,
                              state_dict=synth.synth_state_dict(cfg, seed=0, device=dev))
    pipe = CausalInferencePipeline(bench._pipe_args(), dev, generator=gen)
    noise = synth.synth_noise(cfg, 3 * (4 + a.blocks), seed=0, device=dev)
    st = pipe.stream(noise, {"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1, device=dev)})
    for _ in range(4):
        next(st)
    torch.cuda.synchronize()
    samples = {i: [] for i in range(len(hs))}
    stop = threading.Event()

    def loop():
        while not stop.is_set():
            for i, h in enumerate(hs):
                try:
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                    clk = [c for c in m.get("current_gfxclks", []) if isinstance(c, (int, float)) and 0 < c < 60000]
                    if clk:
                        samples[i].append((clk, m.get("average_socket_power"), m.get("current_socket_power")))
                except Exception as exc:      # noqa: BLE001
                    samples[i].append(repr(exc)[:80])
            stop.wait(0.05)

    from longlive_amd import ops
    bf = torch.bfloat16
    if a.load == "gemm":
        X = torch.randn(1, 4680, 1536, device=dev).to(bf)
        W = (torch.randn(8960, 1536, device=dev) * 0.0255).to(bf)
        bias = torch.zeros(8960, device=dev, dtype=bf)
        work = lambda: ops.gemm(X, W, bias, ops.EPI_BIAS_GELU)
        reps = 250 * a.blocks
    elif a.load == "attn":
        q = torch.randn(1, 4680, 12, 128, device=dev).to(bf)
        k = torch.randn(1, 18720, 12, 128, device=dev).to(bf)
        v = torch.randn(1, 18720, 12, 128, device=dev).to(bf)
        work = lambda: ops.flash_attn(q, k, v, [(0, 18720)])
        reps = 70 * a.blocks
    thr = threading.Thread(target=loop, daemon=True)
    thr.start()
    t0 = time.perf_counter()
    with torch.no_grad():
        if a.load == "pipe":
            for _ in range(a.blocks):
                next(st)
        else:
            for _ in range(reps):
                work()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    thr.join(timeout=2)
    out = {"load": a.load, "seconds": dt, "frames_per_s": 12 * a.blocks / dt if a.load == "pipe" else None, "devices": {}}
    for i, ss in samples.items():
        good = [s for s in ss if isinstance(s, tuple)]
        if not good:
            out["devices"][str(i)] = {"samples": 0, "note": str(ss[:1])}
            continue
        n = len(good[0][0])
        per = [[s[0][k] for s in good if len(s[0]) == n] for k in range(n)]
        mean = [sum(p) / len(p) for p in per]
        busy = max(mean) > 1000
        out["devices"][str(i)] = {"samples": len(good), "xcd_gfxclk_mhz_mean": [round(x, 1) for x in mean],
                                  "xcd_gfxclk_mhz_min": [min(p) for p in per], "xcd_gfxclk_mhz_max": [max(p) for p in per],
                                  "spread_of_means_pct": round(100 * (max(mean) - min(mean)) / max(mean), 2) if busy else None,
                                  "power_w": [s[1] or s[2] for s in good][:: max(1, len(good) // 8)]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
