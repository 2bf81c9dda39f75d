#!/usr/bin/env python3
"""Runs BASELINE.json's long configurations on one MI355X with the pipelines' own profiler and prints one JSON object:
  config 3: 60-s single-prompt generation (T = 240 latent frames, sliding KV cache + frame sink)
  config 4: interactive multi-prompt stream (T = 240, 6 prompts, switches at 40,80,120,160,200, global_sink = false:
            configs/longlive_interactive_inference.yaml:21-27) exercising KV-recache on every switch.
  config 5: `960 --quant int8 --only single`: 240-s generation (T = 960, RoPE frame index < 1024) with W8A8 block linears,
            one replica of the 8 that BASELINE config 5 runs side by side (replicas share nothing: bench.py --gpus 8)
  e2e     : with `--vae` (`--e2e-only` skips configs 3 / 4), the same single-prompt stream with every block decoded live by the HIP VAE decoder
            (pipeline.stream_video): generated pixel frames/s including the decode, and the latency of each block's frames.
Random-init LongLive-1.3B / Wan-VAE weights, synthetic prompt embeddings / noise (longlive_amd.synth)."""
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from longlive_amd import synth  # noqa: E402
from longlive_amd.pipeline import CausalInferencePipeline, InteractiveCausalInferencePipeline  # noqa: E402
from longlive_amd.wan_wrapper import WanDiffusionWrapper  # noqa: E402


def args(gs):
    return SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                           context_noise=0, global_sink=gs)


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    T = int(argv[0]) if argv else 240
    dev = torch.device("cuda", 0)
    cfg = synth.longlive_1_3b(local_attn_size=12, sink_size=3)
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=12, sink_size=3, cfg=cfg, device=dev,
                              state_dict=synth.synth_state_dict(cfg, seed=0, device=dev))
    quant = "int8" if "int8" in sys.argv else None
    if quant:
        gen.model.set_quant(quant)
    only_single = "single" in sys.argv
    prompts = {f"p{i}": {"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1 + i, device=dev)} for i in range(6)}
    enc = lambda text_prompts: prompts[text_prompts[0]]
    noise = synth.synth_noise(cfg, T, seed=0, device=dev)
    out = {"T_latent": T, "pixel_frames": 4 * T, "quant": quant or "bf16"}

    if "--e2e-only" in sys.argv:
        return e2e(T, dev, gen, enc, noise, out)
    P = CausalInferencePipeline(args(True), dev, generator=gen, text_encoder=enc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, lat = P.inference(noise, ["p0"], return_latents=True, profile=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = P.last_profile
    out["config3_single_prompt"] = {
        "wall_s": dt, "fps_overall": 4 * T / dt, "steady_ms_per_latent_frame": pr["ms_per_latent_frame"],
        "fps_steady": 4000.0 / pr["ms_per_latent_frame"], "block0_ms": pr["block_times_ms"][0],
        "finite": bool(torch.isfinite(lat.float()).all()), "latent_std": float(lat.float().std()),
        "end_indices": [P.kv_cache1[0]["global_end_index"], P.kv_cache1[0]["local_end_index"]]}

    if only_single:
        print(json.dumps(out))
        return
    sw = [s for s in (40, 80, 120, 160, 200) if s < T]
    I = InteractiveCausalInferencePipeline(args(False), dev, generator=gen, text_encoder=enc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _, lat = I.inference(noise, text_prompts_list=[[f"p{i}"] for i in range(len(sw) + 1)], switch_frame_indices=sw,
                         return_latents=True, profile=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = I.last_profile
    out["config4_interactive"] = {
        "wall_s": dt, "fps_overall": 4 * T / dt, "steady_ms_per_latent_frame": pr["ms_per_latent_frame"],
        "switch_blocks": pr["switch_blocks"], "switch_block_ms": pr.get("switch_block_ms"),
        "switch_latency_ms": pr.get("switch_latency_ms"), "finite": bool(torch.isfinite(lat.float()).all()),
        "latent_std": float(lat.float().std()),
        "end_indices": [I.kv_cache1[0]["global_end_index"], I.kv_cache1[0]["local_end_index"]]}
    e2e(T, dev, gen, enc, noise, out)


def e2e(T, dev, gen, enc, noise, out):
    if "--vae" in sys.argv:
        from longlive_amd.vae import WanVAEWrapper
        vcfg = synth.VaeConfig()
        vae = WanVAEWrapper(vcfg, device=dev, chunk=3)
        vae.load_state_dict(synth.synth_vae_state_dict(vcfg, seed=5, device=dev))
        P = CausalInferencePipeline(args(True), dev, generator=gen, text_encoder=enc, vae=vae)
        Te = min(T, 60)
        for key, overlap in (("e2e_stream_video", False), ("e2e_stream_video_overlap_decode", True)):
            times, frames = [], 0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _, px in P.stream_video(noise[:, :Te], ["p0"], overlap_decode=overlap):
                if not overlap:
                    torch.cuda.synchronize()
                else:
                    px[0, -1, 0, 0, 0].item()                  # wait for THIS block's pixels only (the next block keeps running)
                times.append(time.perf_counter()); frames += px.shape[1]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            steady = [(b - a) * 1e3 for a, b in zip(times[5:-1], times[6:])]       # full window from block 6 on
            out[key] = {
                "T_latent": Te, "pixel_frames": frames, "wall_s": dt, "fps_overall": frames / dt,
                "steady_ms_per_block": sum(steady) / max(1, len(steady)),
                "fps_steady_with_vae": 12e3 * len(steady) / sum(steady) if steady else None,
                "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
