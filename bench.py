#!/usr/bin/env python3
"""Headline benchmark: generated frames/s at 832x480 for LongLive-1.3B frame-level AR inference on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one autoregressive block of BASELINE.json's config 2: 3 latent frames (= 12 pixel frames at 832x480)
produced by 4 denoising DiT forwards (warped schedule 1000/937.5/833.3/625, re-noised in between) plus the
clean-context forward that refreshes the block's K/V -- 5 full 30-layer forwards, frame-sink (3) + short-window (12)
self-attention over the KV cache.  W warm-up blocks run first (default 4: the window is then full, so every timed
block is a steady-state block with Lk = 18720, roll + insert -- the regime of the reference's 20.7 FPS figure and the
most expensive one), then exactly K blocks are timed between barrier + device sync on both sides.

Multi-GPU = independent replicas (the path does not shard): every rank generates its own stream
(own seed / prompt), no data-path collective; value = frames of all ranks / max-over-ranks time.

The JSON line also carries
  roofline     : the dominant kernel (self-attention flash kernel), timed live with HIP events on its launch stream over
                 the timed region; achieved = algorithmic FLOPs (4 * Lq * Lk * 128 * heads) / avg launch duration
  cpu_baseline : the CPU oracle (a port of the reference's PyTorch path, oracle/) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16
PIXEL_FRAMES_PER_LATENT = 4               # VAE temporal stride (wan/configs/wan_t2v_1_3B.py:17)
BASELINE_FPS = None                       # BASELINE.json "published": {} -> no number for this exact metric on MI355X


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7, help="timed AR blocks (7 blocks = one 5-second clip)")
    ap.add_argument("--warmup", type=int, default=4, help="untimed AR blocks (4 fill the 12-frame window)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--cpu-layers", type=int, default=2, help="layers of one steady-state forward timed on the CPU")
    ap.add_argument("--workload", choices=["dit", "vae", "t5"], default="dit",
                    help="dit (default): the headline metric.  vae / t5: the section-8f rows (VAE decoder, umT5 encoder) with "
                         "their own roofline and cpu_baseline objects; single GPU, not the driver's metric")
    ap.add_argument("--quant", choices=["none", "int8"], default="none",
                    help="int8: W8A8 block linears (BASELINE config 5); the headline metric is the default bf16 path")
    return ap.parse_args()


def cpu_baseline(num_layers_sample: int):
    """Times the CPU oracle on `num_layers_sample` of the 30 layers of ONE steady-state DiT forward (L = 4680 query
    tokens, full 18720-slot KV cache, roll + insert) and extrapolates to a block (x 30/sample layers x 5 forwards)."""
    from longlive_amd import synth
    from oracle import ref_model as RM

    cfg = synth.longlive_1_3b(num_layers=num_layers_sample)
    sd = synth.synth_state_dict(cfg, seed=0, layers=list(range(num_layers_sample)))
    fs = cfg.frame_seqlen
    S = 12 * fs
    m = RM.RefModel(RM.RefConfig.from_cfg(cfg), sd, frame_seqlen_for_max_attn=fs)
    kv = RM.new_kv_cache(1, S, num_layers_sample, 12, 128)
    for i, c in enumerate(kv):
        c["k"] = synth.hash_normal(61, f"kv.{i}.k", (1, S, 12, 128)).to(torch.bfloat16)
        c["v"] = (0.5 * synth.hash_normal(61, f"kv.{i}.v", (1, S, 12, 128))).to(torch.bfloat16)
        c["global_end_index"] = S
        c["local_end_index"] = S
    ca = RM.new_crossattn_cache(1, 512, num_layers_sample, 12, 128)
    x = synth.synth_noise(cfg, 3, seed=0).permute(0, 2, 1, 3, 4)
    prompt = synth.synth_prompt_embeds(cfg, seed=1)
    t = torch.full((1, 3), 625.0)
    # a 1-GPU box's CPU share is 16 cores; more threads than that only oversubscribes the host
    threads = min(torch.get_num_threads(), int(os.environ.get("LONGLIVE_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    with torch.no_grad():
        m.forward(x, t, prompt, kv, ca, current_start=S)
    dt = time.perf_counter() - t0
    fwd = dt * 30.0 / num_layers_sample            # embeddings/head are <0.1% of a forward
    block = 5.0 * fwd
    return dict(value=3 * PIXEL_FRAMES_PER_LATENT / block, unit="frames/s", cores=threads, kind="port",
                sample=f"{num_layers_sample} of 30 layers of one steady-state DiT forward (L=4680, Lk=18720) in "
                       f"{dt:.2f}s on {threads} threads, x{30 // num_layers_sample if 30 % num_layers_sample == 0 else 30 / num_layers_sample:.0f} "
                       f"layers x5 forwards per 12-frame block")


def cpu_baseline_vae(vae, lat):
    """The oracle (oracle/ref_vae.py) decoding the FIRST latent frame (one 480x832 pixel frame, 3.4 TFLOP) on the host."""
    from longlive_amd import synth
    from oracle import ref_vae as RV
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    _, layers = synth.vae_decoder_layout(vae.model.cfg)
    dec = RV.RefVaeDecoder({k: v.cpu() for k, v in vae.model.state_dict().items()}, layers)
    t0 = time.perf_counter()
    ref = RV.decode_to_pixel(dec, lat[:, :1].cpu(), use_cache=False)
    dt = time.perf_counter() - t0
    return {"value": ref.shape[1] / dt, "unit": "pixel frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle decode of the first latent frame (1 pixel frame at 480x832), %.1f s" % dt}


def cpu_baseline_t5(enc, cfg, ids, mask):
    """The oracle (oracle/ref_t5.py) on 2 of the 24 layers at the real widths, extrapolated x12."""
    from oracle import ref_t5 as RT
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sd = {k: v.cpu() for k, v in enc.text_encoder.state_dict().items()
          if not k.startswith("blocks.") or int(k.split(".")[1]) < 2}
    t0 = time.perf_counter()
    RT.text_encoder_forward(ids, mask, sd, 2, cfg.num_heads)
    dt = (time.perf_counter() - t0) * cfg.num_layers / 2
    return {"value": 1e3 * dt, "unit": "ms per prompt", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle on 2 of 24 layers, extrapolated x12"}


def side_workload(args):
    """`--workload vae|t5`: tools/vae_bench.py / tools/t5_bench.py (HIP path, roofline) + the CPU baseline leg, which
    lives here because only bench.py may run the oracle outside tests."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.workload == "vae":
        import vae_bench
        rec, vae, lat = vae_bench.run(9, 2)
        if not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_vae(vae, lat)
    else:
        import t5_bench
        rec, enc, cfg, ids, mask = t5_bench.run(5)
        if not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_t5(enc, cfg, ids, mask)
    print(json.dumps(rec), flush=True)


def main():
    args = parse()
    if args.workload != "dit":
        return side_workload(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from longlive_amd import ops, synth
    from longlive_amd.pipeline import CausalInferencePipeline
    from longlive_amd.wan_wrapper import WanDiffusionWrapper

    for kv in filter(None, os.environ.get("LL_TUNING", "").split(",")):   # kernel A/B only, e.g. LL_TUNING=attn_variant=2
        from longlive_amd import _lib
        k, v = kv.split("=")
        _lib.check(_lib.load().ll_set_tuning(k.encode(), int(v)), "ll_set_tuning")
    cfg = synth.longlive_1_3b(local_attn_size=12, sink_size=3)
    sd = synth.synth_state_dict(cfg, seed=0, device=dev)               # random-init weights of the 1.3B architecture
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=12, sink_size=3, cfg=cfg, device=dev, state_dict=sd)
    del sd
    if args.quant == "int8":
        gen.model.set_quant("int8")
    pargs = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                            denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                            context_noise=0, global_sink=True)
    pipe = CausalInferencePipeline(pargs, dev, generator=gen)
    nblocks = args.warmup + args.steps
    T = 3 * nblocks
    assert T <= 1024, "RoPE frame table has 1024 entries"
    # one independent stream per rank (inference.py:49,146: seed + rank, prompts sharded by rank)
    noise = synth.synth_noise(cfg, T, seed=rank, device=dev)
    prompt = {"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1 + rank, device=dev)}

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    stream = pipe.stream(noise, prompt)
    for _ in range(args.warmup):
        next(stream)
    ktimer = None
    if not args.no_kernel_timer:
        ktimer = ops.KernelTimer(tags=("flash_attn_self",))
    barrier()
    ops.timer = ktimer
    t0 = time.perf_counter()
    for _ in range(args.steps):
        next(stream)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.timer = None
    from longlive_amd.replicas import aggregate_throughput
    frames, elapsed = aggregate_throughput(args.steps * 3 * PIXEL_FRAMES_PER_LATENT, elapsed, device=dev)
    fps = frames / elapsed
    out = {
        "metric": "generated frames/sec (832x480) LongLive-1.3B, frame-sink + short-window attention, 4 denoise steps + clean-context pass",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": (fps / BASELINE_FPS) if BASELINE_FPS else None,
        "dtype": "bf16" if args.quant == "none" else "int8 (W8A8 block linears, bf16 attention/norms)", "data": "synthetic",
        "config": {"workload": "LongLive-1.3B 832x480 (latent 16x60x104), 3-frame AR blocks at steady state: "
                               "Lq=4680, Lk=18720 (sink 3 + window 12 frames), 5 DiT forwards/block, 30 layers, random-init weights",
                   "frames_per_step": 3 * PIXEL_FRAMES_PER_LATENT, "ms_per_latent_frame": 1e3 * elapsed / args.steps / 3,
                   "replicas": world, "parallelism": f"replicas x{world} (no collective on the data path)"},
    }
    if rank == 0:
        roof = None
        if ktimer is not None and "flash_attn_self" in ktimer.records:
            s = ktimer.summary()["flash_attn_self"]
            achieved = s["work_per_launch"] / (s["avg_ms"] * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "attn_pmc.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"bound": "mfma", "kernel": "flash_attn_pipe_kernel<8, 1> (self-attention, ping-pong wave groups, Lq=4680, Lk=18720, 12 heads)", "achieved": achieved,
                    "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_DENSE_PEAK_TFLOPS,
                    "traffic": traffic, "launches": s["launches"], "avg_us": 1e3 * s["avg_ms"],
                    "flop_per_launch": s["work_per_launch"],
                    "share_of_step": s["total_ms"] / (1e3 * elapsed)}
        out["roofline"] = roof
        out["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_layers)
            except Exception as exc:      # the baseline is reporting only; never lose the GPU number over it
                out["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
