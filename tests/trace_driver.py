"""Shared driver: replays the call sequence recorded in tests/golden/toy_trace_*.pt (generated from the
reference by oracle/make_golden.py::_toy_trace) against any backend (the CPU oracle or the HIP product path).

A backend exposes:
    fwd(x[B,F,C,H,W], prompt_embeds, timestep[B,F] f32, current_start, sink_recache) -> x0
    zero_kv(); reset_cross(); indices() -> (G0, E0, G_last, E_last); kv_tensors() -> [(k, v)] per layer
    add_noise(x0_flat, noise_flat, t_flat) -> noisy_flat
"""
import torch

from longlive_amd import synth


class HashRandn:
    def __init__(self, seed):
        self.seed, self.i = seed, 0

    def __call__(self, like):
        x = synth.hash_normal(self.seed, f"renoise.{self.i}", tuple(like.shape)).to(like.dtype)
        self.i += 1
        return x.to(like.device)


def trace_inputs(rec):
    cfg = synth.WanConfig(**rec["cfg"])
    B, T = rec["B"], rec["T"]
    noise = synth.synth_noise(cfg, T, seed=5, batch=B)
    prompts = [synth.synth_prompt_embeds(cfg, seed=7 + i, batch=B, valid_tokens=9) for i in range(3)]
    return cfg, noise, prompts


def replay(rec, backend, noise, prompts, device="cpu"):
    """Returns dict(x0s=[...], idx=[...], caches={...}, output=...) in the golden's structure."""
    cfg = synth.WanConfig(**rec["cfg"])
    fs = cfg.frame_seqlen
    nfb, T, B, steps = rec["nfb"], rec["T"], rec["B"], rec["steps"]
    recache_at = rec["recache_at"]
    rnd = HashRandn(9)
    noise = noise.to(device)
    prompts = [p.to(device) for p in prompts]
    out = torch.zeros_like(noise)
    got = dict(x0s=[], idx=[], caches={})
    seg, start = 0, 0

    def fwd(x, prompt, tval, cs, sink_recache=False):
        t = torch.ones([B, x.shape[1]], dtype=torch.float32, device=device) * tval
        x0 = backend.fwd(x, prompt, t, cs, sink_recache)
        got["x0s"].append(x0.detach().cpu().clone())
        got["idx"].append(tuple(backend.indices()))
        return x0

    for blk in range(T // nfb):
        if start in recache_at:
            global_sink = recache_at[start]
            seg += 1
            if not global_sink:
                backend.zero_kv()
            backend.reset_cross()
            nre = start if cfg.local_attn_size == -1 else min(cfg.local_attn_size, start)
            fwd(out[:, start - nre:start], prompts[seg], 0.0, (start - nre) * fs, sink_recache=not global_sink)
            backend.reset_cross()
            got["caches"][f"after_recache_{start}"] = [(k.cpu().clone(), v.cpu().clone()) for k, v in backend.kv_tensors()]
        noisy = noise[:, start:start + nfb]
        for i, tv in enumerate(steps):
            x0 = fwd(noisy, prompts[seg], tv, start * fs)
            if i < len(steps) - 1:
                tn = steps[i + 1] * torch.ones([B * nfb], device=device)
                noisy = backend.add_noise(x0.flatten(0, 1), rnd(x0.flatten(0, 1)), tn).unflatten(0, x0.shape[:2])
        out[:, start:start + nfb] = x0
        fwd(x0, prompts[seg], 0.0, start * fs)
        key = f"after_block_{blk}"
        if key in rec["caches"]:
            got["caches"][key] = [(k.cpu().clone(), v.cpu().clone()) for k, v in backend.kv_tensors()]
        start += nfb
    got["output"] = out.cpu()
    return got
