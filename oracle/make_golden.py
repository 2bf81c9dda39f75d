"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.pt by running the REFERENCE's own modules on CPU.

Run by hand in the build container (the only place /root/reference exists):

    python -m oracle.make_golden [ops toy pipe real_block real_fwd]

Inputs and weights come from longlive_amd.synth (integer counter hash), so only seeds + outputs are
stored and any host / the MI355X can regenerate bit-identical inputs without the reference.
The reference draws re-noise with torch.randn_like (pipeline/causal_inference.py:175); while a pipeline
golden is generated that call is answered from the same counter hash (`_HashRandn`) so that the draw is
reproducible elsewhere.
"""
from __future__ import annotations

import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.nn as nn

from longlive_amd import synth
from oracle.ref_import import load_pipelines

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
torch.set_grad_enabled(False)


def _save(name, obj):
    os.makedirs(OUT, exist_ok=True)
    p = os.path.join(OUT, name)
    torch.save(obj, p)
    print(f"wrote {p}  ({os.path.getsize(p) / 1e6:.2f} MB)")


class _HashRandn:
    """Replacement for torch.randn_like inside reference pipelines: call i returns hash_normal(seed, 'renoise.i')."""

    def __init__(self, seed):
        self.seed, self.i = seed, 0

    def __call__(self, like, **kw):
        x = synth.hash_normal(self.seed, f"renoise.{self.i}", tuple(like.shape)).to(like.dtype)
        self.i += 1
        return x


def build_ref_model(ns, cfg: synth.WanConfig, sd, frame_seqlen):
    M = ns.causal_model.CausalWanModel(
        model_type="t2v", patch_size=cfg.patch_size, text_len=cfg.text_len, in_dim=cfg.in_dim, dim=cfg.dim,
        ffn_dim=cfg.ffn_dim, freq_dim=cfg.freq_dim, text_dim=cfg.text_dim, out_dim=cfg.out_dim,
        num_heads=cfg.num_heads, num_layers=cfg.num_layers, local_attn_size=cfg.local_attn_size,
        sink_size=cfg.sink_size, qk_norm=True, cross_attn_norm=True, eps=cfg.eps)
    missing = M.load_state_dict(sd, strict=True)
    M = M.to(torch.bfloat16).eval()
    # what CausalInferencePipeline._set_all_modules_max_attention_size does (causal_inference.py:295-329)
    tgt = 32760 if cfg.local_attn_size == -1 else cfg.local_attn_size * frame_seqlen
    M.max_attention_size = tgt
    for m in M.modules():
        if hasattr(m, "max_attention_size"):
            m.max_attention_size = tgt
    M.local_attn_size = cfg.local_attn_size
    return M


def build_ref_wrapper(ns, model, shift=5.0):
    W = ns.wan_wrapper.WanDiffusionWrapper
    g = W.__new__(W)
    nn.Module.__init__(g)
    g.model = model
    g.uniform_timestep = False
    g.scheduler = ns.scheduler.FlowMatchScheduler(shift=shift, sigma_min=0.0, extra_one_step=True)
    g.scheduler.set_timesteps(1000, training=True)
    g.seq_len = 32760
    g.post_init()
    return g


def ref_kv_cache(B, S, L, n, d):
    return [dict(k=torch.zeros(B, S, n, d, dtype=torch.bfloat16), v=torch.zeros(B, S, n, d, dtype=torch.bfloat16),
                 global_end_index=torch.tensor([0]), local_end_index=torch.tensor([0])) for _ in range(L)]


def ref_ca_cache(B, T, L, n, d):
    return [dict(k=torch.zeros(B, T, n, d, dtype=torch.bfloat16), v=torch.zeros(B, T, n, d, dtype=torch.bfloat16),
                 is_init=False) for _ in range(L)]


# --------------------------------------------------------------------------------------------
def gen_ops(ns):
    g = {}
    bf = torch.bfloat16
    dim, eps = 256, 1e-6
    x = (synth.hash_normal(11, "ops.x", (2, 48, dim)) * 1.7 + 0.3).to(bf)
    w = (1 + 0.1 * synth.hash_normal(11, "ops.w", (dim,))).to(bf)
    b = (0.1 * synth.hash_normal(11, "ops.b", (dim,))).to(bf)
    rn = ns.model.WanRMSNorm(dim, eps=eps); rn.weight.data = w.clone(); rn = rn.to(bf)
    g["rms_norm"] = rn(x)
    ln = ns.model.WanLayerNorm(dim, eps); g["layer_norm"] = ln(x)
    la = ns.model.WanLayerNorm(dim, eps, elementwise_affine=True).to(bf)
    la.weight.data = w.clone(); la.bias.data = b.clone()
    g["layer_norm_affine"] = la(x)
    e = (synth.hash_normal(11, "ops.e", (2, 2, 6, dim)) * 0.5).to(bf)
    ec = e.chunk(6, dim=2)
    g["ln_modulate"] = (ln(x).unflatten(1, (2, 24)) * (1 + ec[1]) + ec[0]).flatten(1, 2)   # causal_model.py:445
    y = (synth.hash_normal(11, "ops.y", (2, 48, dim))).to(bf)
    g["gate_residual"] = x + (y.unflatten(1, (2, 24)) * ec[2]).flatten(1, 2)               # causal_model.py:456
    # rope
    q = synth.hash_normal(11, "ops.q", (2, 48, 2, 128)).to(bf)
    freqs = torch.cat([ns.model.rope_params(1024, 44), ns.model.rope_params(1024, 42), ns.model.rope_params(1024, 42)], dim=1)
    grid = torch.tensor([[2, 4, 6], [2, 4, 6]])
    for sf in (0, 5, 959):
        g[f"rope_sf{sf}"] = ns.causal_model.causal_rope_apply(q, grid, freqs, start_frame=sf).type_as(q)
    g["freqs_real"] = torch.view_as_real(freqs).to(torch.float64)[:8].clone()
    # time embedding sinusoid
    t = torch.tensor([1000.0, 937.5, 833.3333, 625.0, 0.0, 3.0])
    g["sinusoid"] = ns.model.sinusoidal_embedding_1d(256, t)
    # scheduler
    sch = ns.scheduler.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(1000, training=True)
    g["sigmas"] = sch.sigmas.clone(); g["timesteps"] = sch.timesteps.clone()
    ts = torch.cat((sch.timesteps, torch.tensor([0.0])))[1000 - torch.tensor([1000, 750, 500, 250])]
    g["warped_steps"] = ts
    x0 = synth.hash_normal(11, "ops.x0", (4, 16, 8, 12)).to(bf)
    nz = synth.hash_normal(11, "ops.nz", (4, 16, 8, 12)).to(bf)
    g["add_noise"] = sch.add_noise(x0, nz, ts[[1, 2, 3, 3]] * torch.ones(4, dtype=torch.long))
    wr = build_ref_wrapper(ns, nn.Identity())
    g["flow_to_x0"] = wr._convert_flow_pred_to_x0(nz, x0, ts)
    # attention (CPU SDPA fallback, bf16)
    qa = synth.hash_normal(11, "ops.qa", (1, 40, 2, 128)).to(bf)
    ka = synth.hash_normal(11, "ops.ka", (1, 72, 2, 128)).to(bf)
    va = synth.hash_normal(11, "ops.va", (1, 72, 2, 128)).to(bf)
    g["attention"] = ns.attention.attention(qa, ka, va)
    _save("ops.pt", g)


# --------------------------------------------------------------------------------------------
def _toy_trace(ns, name, cfg: synth.WanConfig, nfb: int, T: int, B: int, recache_at, steps=(1000.0, 700.0), keep_blocks=(0, 3)):
    """Drive reference CausalWanModel._forward_inference through the pipeline's call sequence by hand
    (the reference pipelines hard-code 12x128 heads, so the 2-head toy cannot go through them)."""
    fs = cfg.frame_seqlen
    sd = synth.synth_state_dict(cfg, seed=3)
    M = build_ref_model(ns, cfg, {k: v.float() for k, v in sd.items()}, fs)
    wr = build_ref_wrapper(ns, M)
    S = (cfg.local_attn_size if cfg.local_attn_size != -1 else T) * fs
    kv = ref_kv_cache(B, S, cfg.num_layers, cfg.num_heads, cfg.head_dim)
    ca = ref_ca_cache(B, cfg.text_len, cfg.num_layers, cfg.num_heads, cfg.head_dim)
    noise = synth.synth_noise(cfg, T, seed=5, batch=B)
    prompts = [synth.synth_prompt_embeds(cfg, seed=7 + i, batch=B, valid_tokens=9) for i in range(3)]
    rnd = _HashRandn(9)
    out = torch.zeros_like(noise)
    rec = dict(x0s=[], idx=[], caches={}, cfg=vars(cfg).copy(), nfb=nfb, T=T, B=B,
               recache_at=dict(recache_at), steps=list(steps))
    seg, start = 0, 0

    def fwd(x, prompt, tval, cs, sink_recache=False):
        t = torch.ones([B, x.shape[1]], dtype=torch.float32) * tval
        flow, x0 = wr(noisy_image_or_video=x, conditional_dict={"prompt_embeds": prompt}, timestep=t,
                      kv_cache=kv, crossattn_cache=ca, current_start=cs, sink_recache_after_switch=sink_recache)
        rec["x0s"].append(x0.clone())
        rec["idx"].append((int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"]),
                           int(kv[-1]["global_end_index"]), int(kv[-1]["local_end_index"])))
        return x0

    for blk in range(T // nfb):
        if start in recache_at:
            global_sink = recache_at[start]
            seg += 1
            if not global_sink:
                for c in kv:
                    c["k"].zero_(); c["v"].zero_()
            for c in ca:
                c["k"].zero_(); c["v"].zero_(); c["is_init"] = False
            nre = start if cfg.local_attn_size == -1 else min(cfg.local_attn_size, start)
            fwd(out[:, start - nre:start], prompts[seg], 0.0, (start - nre) * fs, sink_recache=not global_sink)
            for c in ca:
                c["k"].zero_(); c["v"].zero_(); c["is_init"] = False
            rec["caches"][f"after_recache_{start}"] = [(c["k"].clone(), c["v"].clone()) for c in kv]
        noisy = noise[:, start:start + nfb]
        for i, tv in enumerate(steps):
            x0 = fwd(noisy, prompts[seg], tv, start * fs)
            if i < len(steps) - 1:
                noisy = wr.scheduler.add_noise(x0.flatten(0, 1), rnd(x0.flatten(0, 1)),
                                               steps[i + 1] * torch.ones([B * nfb])).unflatten(0, x0.shape[:2])
        out[:, start:start + nfb] = x0
        fwd(x0, prompts[seg], 0.0, start * fs)
        if blk in keep_blocks or blk == T // nfb - 1:
            rec["caches"][f"after_block_{blk}"] = [(c["k"].clone(), c["v"].clone()) for c in kv]
        start += nfb
    rec["output"] = out
    _save(name, rec)


def gen_toy(ns):
    _toy_trace(ns, "toy_trace_f1_w3_s1.pt", synth.toy_config(local_attn_size=3, sink_size=1), nfb=1, T=9, B=2,
               recache_at={5: False, 7: True})
    _toy_trace(ns, "toy_trace_f2_w5_s2.pt", synth.toy_config(local_attn_size=5, sink_size=2), nfb=2, T=12, B=1,
               recache_at={8: False})
    _toy_trace(ns, "toy_trace_f1_w4_s0.pt", synth.toy_config(local_attn_size=4, sink_size=0), nfb=1, T=7, B=1,
               recache_at={})
    _toy_trace(ns, "toy_trace_global.pt", synth.toy_config(local_attn_size=-1, sink_size=0), nfb=1, T=4, B=1,
               recache_at={})


# --------------------------------------------------------------------------------------------
def pipe_cfg():
    """Real width/heads/ffn/text (the reference pipelines hard-code 12x128 heads), 2 layers, 8x12 latents."""
    return synth.WanConfig(num_layers=2, lat_h=8, lat_w=12, local_attn_size=12, sink_size=3)


def _fake_text_encoder(prompt_table):
    def enc(text_prompts):
        return {"prompt_embeds": prompt_table[text_prompts[0]]}
    return enc


class _FakeVAE:
    def decode_to_pixel(self, x, use_cache=False):
        return x


def gen_pipe(ns):
    cfg = pipe_cfg()
    fs = cfg.frame_seqlen
    sd = synth.synth_state_dict(cfg, seed=21)
    M = build_ref_model(ns, cfg, {k: v.float() for k, v in sd.items()}, fs)
    wr = build_ref_wrapper(ns, M)
    table = {f"p{i}": synth.synth_prompt_embeds(cfg, seed=31 + i) for i in range(3)}
    args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                           num_frame_per_block=3, context_noise=0, global_sink=True)
    real_randn_like = torch.randn_like
    rec = dict(cfg=vars(cfg).copy())
    try:
        # config-2-like single prompt, T = 21
        P = ns.causal_inference.CausalInferencePipeline(args, "cpu", generator=wr, text_encoder=_fake_text_encoder(table),
                                                        vae=_FakeVAE())
        P.num_transformer_blocks, P.frame_seq_length = cfg.num_layers, fs
        noise = synth.synth_noise(cfg, 21, seed=41)
        torch.randn_like = _HashRandn(43)
        _, lat = P.inference(noise, ["p0"], return_latents=True)
        rec["single_T21"] = lat.clone()
        rec["single_T21_idx"] = (int(P.kv_cache1[0]["global_end_index"]), int(P.kv_cache1[0]["local_end_index"]))
        # interactive, T = 24, switches 7 / 16, with and without global sink
        for gs in (False, True):
            args.global_sink = gs
            I = ns.interactive.InteractiveCausalInferencePipeline(args, "cpu", generator=wr,
                                                                  text_encoder=_fake_text_encoder(table), vae=_FakeVAE())
            I.num_transformer_blocks, I.frame_seq_length = cfg.num_layers, fs
            noise = synth.synth_noise(cfg, 24, seed=45)
            torch.randn_like = _HashRandn(47)
            _, lat = I.inference(noise, text_prompts_list=[["p0"], ["p1"], ["p2"]], switch_frame_indices=[7, 16],
                                 return_latents=True)
            rec[f"interactive_T24_gs{int(gs)}"] = lat.clone()
    finally:
        torch.randn_like = real_randn_like
    _save("pipe_w1536_l2.pt", rec)


# --------------------------------------------------------------------------------------------
def synth_kv_fill(cfg, layer, S, fill, seed=61):
    """Synthetic steady-state cache content: slots [0, fill) ~ N(0,1) keys / 0.5 N(0,1) values."""
    k = torch.zeros(1, S, cfg.num_heads, cfg.head_dim, dtype=torch.bfloat16)
    v = torch.zeros_like(k)
    k[:, :fill] = synth.hash_normal(seed, f"kv.{layer}.k", (1, fill, cfg.num_heads, cfg.head_dim)).to(torch.bfloat16)
    v[:, :fill] = (0.5 * synth.hash_normal(seed, f"kv.{layer}.v", (1, fill, cfg.num_heads, cfg.head_dim))).to(torch.bfloat16)
    return k, v


def sample_rows(L, n=48):
    return torch.linspace(0, L - 1, n).round().long()


def gen_real_block(ns):
    """One CausalWanAttentionBlock at the real 1.3B shape, steady state: full 18720-slot cache, roll + insert."""
    cfg = synth.longlive_1_3b(num_layers=1)
    fs = cfg.frame_seqlen
    sd = synth.synth_state_dict(cfg, seed=0, layers=[0])
    M = build_ref_model(ns, cfg, {k: v.float() for k, v in sd.items()}, fs)
    blk = M.blocks[0]
    S = 12 * fs
    x = synth.hash_normal(71, "blk.x", (1, 3 * fs, cfg.dim)).to(torch.bfloat16)
    e0 = (0.3 * synth.hash_normal(71, "blk.e0", (1, 3, 6, cfg.dim))).to(torch.bfloat16)
    ctx = synth.hash_normal(71, "blk.ctx", (1, cfg.text_len, cfg.dim)).to(torch.bfloat16)
    k, v = synth_kv_fill(cfg, 0, S, S)
    kv = dict(k=k, v=v, global_end_index=torch.tensor([S]), local_end_index=torch.tensor([S]))
    ca = dict(k=torch.zeros(1, 512, 12, 128, dtype=torch.bfloat16), v=torch.zeros(1, 512, 12, 128, dtype=torch.bfloat16),
              is_init=False)
    grid = torch.tensor([[3, 30, 52]])
    t0 = time.time()
    y, (cur_end, local_end, info) = blk(x, e0, torch.tensor([3 * fs]), grid, M.freqs, ctx, None, None,
                                        kv_cache=kv, crossattn_cache=ca, current_start=S)
    print(f"real block: {time.time() - t0:.1f}s")
    M._apply_cache_updates([kv], [(0, (cur_end, local_end, info))])
    rows = sample_rows(3 * fs)
    slots = sample_rows(S, 64)
    _save("real_block.pt", dict(rows=rows, y_rows=y[0, rows].clone(), slots=slots,
                                k_slots=kv["k"][0, slots].clone(), v_slots=kv["v"][0, slots].clone(),
                                idx=(int(kv["global_end_index"]), int(kv["local_end_index"])),
                                y_mean=y.float().mean().item(), y_std=y.float().std().item()))


def gen_real_fwd(ns):
    """Full 30-layer forward at the real shape: block 0 (Lk = 4680), t = 1000, then the same weights at steady
    state (synthetic full caches, roll path, Lk = 18720), t = 625."""
    cfg = synth.longlive_1_3b()
    fs = cfg.frame_seqlen
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, seed=0)
    print(f"weights: {time.time() - t0:.1f}s")
    M = build_ref_model(ns, cfg, sd, fs)   # load_state_dict copies bf16 -> fp32 params, then .to(bf16): exact
    wr = build_ref_wrapper(ns, M)
    S = 12 * fs
    prompt = synth.synth_prompt_embeds(cfg, seed=1)
    noise = synth.synth_noise(cfg, 3, seed=0)
    rec = {}
    kv = ref_kv_cache(1, S, 30, 12, 128); ca = ref_ca_cache(1, 512, 30, 12, 128)
    t0 = time.time()
    flow, x0 = wr(noise, {"prompt_embeds": prompt}, torch.ones(1, 3) * 1000.0, kv_cache=kv, crossattn_cache=ca,
                  current_start=0)
    print(f"real fwd block0: {time.time() - t0:.1f}s")
    rec["flow_block0"] = flow.clone(); rec["x0_block0"] = x0.clone()
    slots = sample_rows(3 * fs, 32)
    rec["slots0"] = slots
    rec["k_l0_block0"] = kv[0]["k"][0, slots].clone(); rec["k_l29_block0"] = kv[29]["k"][0, slots].clone()
    rec["v_l29_block0"] = kv[29]["v"][0, slots].clone()
    # steady state
    for i in range(30):
        k, v = synth_kv_fill(cfg, i, S, S)
        kv[i]["k"], kv[i]["v"] = k, v
        kv[i]["global_end_index"].fill_(S); kv[i]["local_end_index"].fill_(S)
    t0 = time.time()
    flow, x0 = wr(noise, {"prompt_embeds": prompt}, torch.ones(1, 3) * 625.0, kv_cache=kv, crossattn_cache=ca,
                  current_start=S)
    print(f"real fwd steady: {time.time() - t0:.1f}s")
    rec["flow_steady"] = flow.clone(); rec["x0_steady"] = x0.clone()
    slots = sample_rows(S, 64)
    rec["slots_steady"] = slots
    rec["k_l29_steady"] = kv[29]["k"][0, slots].clone()
    rec["idx_steady"] = (int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"]))
    _save("real_fwd.pt", rec)


def gen_config1(ns):
    """BASELINE config 1 (SURVEY section 8c "G5"): LongLive-1.3B shape (30 layers, 832x480 latents), random-init,
    `denoising_step_list=[1000]`, a 4-latent-frame clip through the REFERENCE's CausalInferencePipeline on CPU in bf16
    (pipeline/causal_inference.py:56-253).  4 % 3 != 0 (causal_inference.py:77), so num_frame_per_block = 1: four blocks
    of (1 denoise forward + 1 clean-context forward).  One denoise step => no randn_like draw."""
    cfg = synth.longlive_1_3b()
    fs = cfg.frame_seqlen
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, seed=0)
    M = build_ref_model(ns, cfg, sd, fs)
    del sd
    wr = build_ref_wrapper(ns, M)
    print(f"config1 model: {time.time() - t0:.1f}s")
    table = {"p0": synth.synth_prompt_embeds(cfg, seed=1)}
    args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000], warp_denoising_step=True, num_frame_per_block=1,
                           context_noise=0, global_sink=True)
    P = ns.causal_inference.CausalInferencePipeline(args, "cpu", generator=wr, text_encoder=_fake_text_encoder(table),
                                                    vae=_FakeVAE())
    assert (P.num_transformer_blocks, P.frame_seq_length) == (30, fs)
    noise = synth.synth_noise(cfg, 4, seed=0)
    t0 = time.time()
    _, lat = P.inference(noise, ["p0"], return_latents=True)
    print(f"config1 pipeline (4 frames, 8 forwards): {time.time() - t0:.1f}s")
    slots = sample_rows(4 * fs, 64)
    _save("config1_pipe.pt", dict(latents=lat.clone(), slots=slots,
                                  k_l0=P.kv_cache1[0]["k"][0, slots].clone(), v_l0=P.kv_cache1[0]["v"][0, slots].clone(),
                                  k_l29=P.kv_cache1[29]["k"][0, slots].clone(), v_l29=P.kv_cache1[29]["v"][0, slots].clone(),
                                  idx=(int(P.kv_cache1[0]["global_end_index"]), int(P.kv_cache1[0]["local_end_index"])),
                                  kv_shape=list(P.kv_cache1[0]["k"].shape)))


def gen_real_recache(ns):
    """BASELINE config 4's prompt-switch forward at the real shape, 2 layers: the reference's
    InteractiveCausalInferencePipeline._recache_after_switch (interactive_causal_inference.py:34-106) re-encodes the last
    12 frames in ONE forward (L = Lk = 18720) under the new prompt -- with global_sink False (caches zeroed first,
    sink_recache_after_switch=True) and True (caches kept, sink slots protected).  State before the switch: window full
    (synthetic cache content, end indices (24 frames, 12 frames)); `output` holds 24 synthetic clean latent frames."""
    cfg = synth.longlive_1_3b(num_layers=2)
    fs = cfg.frame_seqlen
    S = 12 * fs
    sd = synth.synth_state_dict(cfg, seed=0, layers=[0, 1])
    M = build_ref_model(ns, cfg, {k: v.float() for k, v in sd.items()}, fs)
    wr = build_ref_wrapper(ns, M)
    table = {"p1": synth.synth_prompt_embeds(cfg, seed=2)}
    output = synth.synth_noise(cfg, 24, seed=7)          # stands in for the clean latents generated so far
    rec = dict(slots=sample_rows(S, 96), frames=[0, 5, 11])
    for gs in (False, True):
        args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                               denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                               context_noise=0, global_sink=gs)
        I = ns.interactive.InteractiveCausalInferencePipeline(args, "cpu", generator=wr,
                                                              text_encoder=_fake_text_encoder(table), vae=_FakeVAE())
        I.num_transformer_blocks = 2
        I.kv_cache1 = ref_kv_cache(1, S, 2, 12, 128)
        I.crossattn_cache = ref_ca_cache(1, 512, 2, 12, 128)
        for i, c in enumerate(I.kv_cache1):
            c["k"], c["v"] = synth_kv_fill(cfg, i, S, S)
            c["global_end_index"].fill_(24 * fs); c["local_end_index"].fill_(S)
        I._set_all_modules_max_attention_size(12)
        x0s = []
        orig_fwd = wr.forward

        def spy(*a, **k):
            out = orig_fwd(*a, **k)
            x0s.append(out[1].clone())
            return out
        wr.forward = spy
        t0 = time.time()
        try:
            I._recache_after_switch(output, 24, {"prompt_embeds": table["p1"]})
        finally:
            wr.forward = orig_fwd
        print(f"real recache gs={gs}: {time.time() - t0:.1f}s")
        assert len(x0s) == 1 and x0s[0].shape[1] == 12
        tag = f"gs{int(gs)}"
        rec[tag] = dict(x0_frames=x0s[0][:, rec["frames"]].clone(),
                        k=[c["k"][0, rec["slots"]].clone() for c in I.kv_cache1],
                        v=[c["v"][0, rec["slots"]].clone() for c in I.kv_cache1],
                        idx=(int(I.kv_cache1[0]["global_end_index"]), int(I.kv_cache1[0]["local_end_index"])),
                        ca_init=[bool(c["is_init"]) for c in I.crossattn_cache])
    _save("real_recache.pt", rec)


CONFIG2_SLOTS = 24            # cache slots sampled per (block, layer, k|v)
CONFIG2_LAYERS = (0, 14, 29)
CONFIG2_FULL_STEP_BLOCKS = (0, 4, 6)   # blocks whose three intermediate x0 are stored whole (teacher-forced per step)
CONFIG2_NSAMPLE = 8192


def gen_config2(ns):
    """BASELINE config 2 exactly, through the REFERENCE's CausalInferencePipeline on CPU in bf16
    (pipeline/causal_inference.py:56-253): LongLive-1.3B shape (30 layers, 60x104 latents), T = 21 latent frames,
    denoising_step_list [1000, 750, 500, 250] (warped), num_frame_per_block 3, window 12 / sink 3, context_noise 0:
    7 blocks x (4 denoise forwards + re-noise + clean-context forward) = 35 forwards; the window fills during blocks
    0-3 and ROLLS in blocks 4-6, every cache entry written by the model itself.  Re-noise draws come from `_HashRandn`.

    Stored: the 21 latent frames; x0 of every denoise forward (whole for blocks 0 / 4 / 6, a fixed 8192-element sample
    for the others); after each block's context pass 24 sampled slots of K and V of layers 0 / 14 / 29 + end indices.

    Then config 4's switch at full depth on THAT state: InteractiveCausalInferencePipeline._recache_after_switch
    (interactive_causal_inference.py:34-106) re-encodes frames 9..20 under a new prompt in ONE 30-layer forward
    (L = Lk = 18720), with global_sink True (caches kept, sink protected) and False (caches zeroed, sink rewritten):
    x0 of three frames + 96 sampled slots of K / V of layers 0 / 14 / 29 + end indices."""
    import copy
    cfg = synth.longlive_1_3b()
    fs = cfg.frame_seqlen
    T, nfb = 21, 3
    t0 = time.time()
    sd = synth.synth_state_dict(cfg, seed=0)
    M = build_ref_model(ns, cfg, sd, fs)
    del sd
    wr = build_ref_wrapper(ns, M)
    print(f"config2 model: {time.time() - t0:.1f}s", flush=True)
    table = {"p0": synth.synth_prompt_embeds(cfg, seed=1), "p1": synth.synth_prompt_embeds(cfg, seed=2)}
    args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=nfb,
                           context_noise=0, global_sink=True)
    P = ns.causal_inference.CausalInferencePipeline(args, "cpu", generator=wr, text_encoder=_fake_text_encoder(table),
                                                    vae=_FakeVAE())
    assert (P.num_transformer_blocks, P.frame_seq_length) == (30, fs)
    noise = synth.synth_noise(cfg, T, seed=0)
    S = 12 * fs
    slots = sample_rows(S, CONFIG2_SLOTS)
    samp = torch.linspace(0, nfb * 16 * cfg.lat_h * cfg.lat_w - 1, CONFIG2_NSAMPLE).round().long()
    rec = dict(T=T, steps=[float(t) for t in P.denoising_step_list], slots=slots, sample_idx=samp, layers=list(CONFIG2_LAYERS),
               renoise_seed=43, noise_seed=0, prompt_seed=1, blocks=[])
    calls = []
    orig_fwd = wr.forward

    def spy(*a, **k):
        t1 = time.time()
        out = orig_fwd(*a, **k)
        n = len(calls)
        blk, j = divmod(n, 5)
        calls.append(dict(t=float(k["timestep"].flatten()[0]), current_start=int(k["current_start"])))
        if j == 0:
            rec["blocks"].append(dict(x0_steps=[], x0_samples=[], flow_std=[]))
        b = rec["blocks"][blk]
        if j < 4:
            x0 = out[1]
            b["x0_samples"].append(x0.flatten()[samp].clone())
            b["flow_std"].append(float(out[0].float().std()))
            if blk in CONFIG2_FULL_STEP_BLOCKS and j < 3:
                b["x0_steps"].append(x0.clone())
        else:
            kv = k["kv_cache"]
            b["k"] = [kv[l]["k"][0, slots].clone() for l in CONFIG2_LAYERS]
            b["v"] = [kv[l]["v"][0, slots].clone() for l in CONFIG2_LAYERS]
            b["idx"] = (int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"]))
        print(f"  call {n} (block {blk}, {'ctx' if j == 4 else 'step %d' % j}, t={calls[-1]['t']:.0f}): "
              f"{time.time() - t1:.1f}s", flush=True)
        return out

    real_randn_like = torch.randn_like
    wr.forward = spy
    try:
        torch.randn_like = _HashRandn(43)
        t0 = time.time()
        _, lat = P.inference(noise, ["p0"], return_latents=True)
        print(f"config2 pipeline (21 frames, {len(calls)} forwards): {time.time() - t0:.1f}s", flush=True)
    finally:
        torch.randn_like = real_randn_like
        wr.forward = orig_fwd
    assert len(calls) == 35
    rec["calls"] = calls
    rec["latents"] = lat.clone()
    _save("config2_pipe.pt", rec)

    # ---- config 4's switch at full depth, entered from the state the 21-frame run left
    rslots = sample_rows(S, 96)
    rr = dict(slots=rslots, frames=[0, 5, 11], layers=list(CONFIG2_LAYERS), start_frame=T, prompt_seed=2)
    for gs in (True, False):
        args.global_sink = gs
        I = ns.interactive.InteractiveCausalInferencePipeline(args, "cpu", generator=wr,
                                                              text_encoder=_fake_text_encoder(table), vae=_FakeVAE())
        I.kv_cache1 = copy.deepcopy(P.kv_cache1) if gs else P.kv_cache1     # the second call may consume the original
        I.crossattn_cache = ref_ca_cache(1, 512, 30, 12, 128)
        I._set_all_modules_max_attention_size(12)
        x0s = []

        def spy2(*a, **k):
            out = orig_fwd(*a, **k)
            x0s.append(out[1].clone())
            return out
        wr.forward = spy2
        t0 = time.time()
        try:
            I._recache_after_switch(lat, T, {"prompt_embeds": table["p1"]})
        finally:
            wr.forward = orig_fwd
        print(f"config2 recache gs={gs}: {time.time() - t0:.1f}s", flush=True)
        assert len(x0s) == 1 and x0s[0].shape[1] == 12
        rr[f"gs{int(gs)}"] = dict(x0_frames=x0s[0][:, rr["frames"]].clone(),
                                  k=[I.kv_cache1[l]["k"][0, rslots].clone() for l in CONFIG2_LAYERS],
                                  v=[I.kv_cache1[l]["v"][0, rslots].clone() for l in CONFIG2_LAYERS],
                                  idx=(int(I.kv_cache1[0]["global_end_index"]), int(I.kv_cache1[0]["local_end_index"])))
        del I
    _save("config2_recache.pt", rr)


def gen_config4(ns):
    """BASELINE config 4 end to end at full depth: the REFERENCE's InteractiveCausalInferencePipeline.inference
    (pipeline/interactive_causal_inference.py:108-432) at 30 layers, 60x104, T = 21, two prompts, switch at frame 12, `global_sink`
    False (configs/longlive_interactive_inference.yaml): blocks 0-3 under p0 (identical to config 2's: same noise, same re-noise
    draws), then `_recache_after_switch` (caches zeroed, ONE 12-frame forward under p1, sink rewritten), then blocks 4-6 under p1 on
    the recached window (which rolls).  36 generator calls.  Stored: latents of frames 12..20, an 8192-element sample of every x0 after
    the switch, x0 of three frames of the recache forward, 24 sampled K / V slots of layers 0 / 14 / 29 + end indices after the recache
    and after every later context pass."""
    cfg = synth.longlive_1_3b()
    fs = cfg.frame_seqlen
    T, nfb, SW = 21, 3, 12
    sd = synth.synth_state_dict(cfg, seed=0)
    M = build_ref_model(ns, cfg, sd, fs)
    del sd
    wr = build_ref_wrapper(ns, M)
    table = {"p0": synth.synth_prompt_embeds(cfg, seed=1), "p1": synth.synth_prompt_embeds(cfg, seed=2)}
    args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=nfb,
                           context_noise=0, global_sink=False)
    I = ns.interactive.InteractiveCausalInferencePipeline(args, "cpu", generator=wr, text_encoder=_fake_text_encoder(table), vae=_FakeVAE())
    assert (I.num_transformer_blocks, I.frame_seq_length) == (30, fs)
    noise = synth.synth_noise(cfg, T, seed=0)
    S = 12 * fs
    slots = sample_rows(S, CONFIG2_SLOTS)
    samp = torch.linspace(0, nfb * 16 * cfg.lat_h * cfg.lat_w - 1, CONFIG2_NSAMPLE).round().long()
    rec = dict(T=T, switch_frame=SW, slots=slots, sample_idx=samp, layers=list(CONFIG2_LAYERS), renoise_seed=43, noise_seed=0,
               prompt_seeds=[1, 2], recache_frames=[0, 5, 11], calls=[], after=[])
    orig_fwd = wr.forward

    def kv_sample(kv):
        return dict(k=[kv[l]["k"][0, slots].clone() for l in CONFIG2_LAYERS], v=[kv[l]["v"][0, slots].clone() for l in CONFIG2_LAYERS],
                    idx=(int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"])))

    def spy(*a, **k):
        t1 = time.time()
        out = orig_fwd(*a, **k)
        n = len(rec["calls"])
        nf = k["noisy_image_or_video"].shape[1]
        c = dict(t=float(k["timestep"].flatten()[0]), current_start=int(k["current_start"]), frames=nf,
                 recache=bool(k.get("sink_recache_after_switch", False)))
        if nf == 12:                                   # the recache forward
            c["x0_frames"] = out[1][:, rec["recache_frames"]].clone()
            c["kv"] = kv_sample(k["kv_cache"])
        elif c["current_start"] >= SW * fs:
            c["x0_sample"] = out[1].flatten()[samp].clone()
            if c["t"] == 0.0:
                c["kv"] = kv_sample(k["kv_cache"])
        rec["calls"].append(c)
        print(f"  call {n} (start {c['current_start'] // fs}, {nf} frames, t={c['t']:.0f}): {time.time() - t1:.1f}s", flush=True)
        return out

    real_randn_like = torch.randn_like
    wr.forward = spy
    try:
        torch.randn_like = _HashRandn(43)
        t0 = time.time()
        _, lat = I.inference(noise, text_prompts_list=[["p0"], ["p1"]], switch_frame_indices=[SW], return_latents=True)
        print(f"config4 pipeline ({len(rec['calls'])} forwards): {time.time() - t0:.1f}s", flush=True)
    finally:
        torch.randn_like = real_randn_like
        wr.forward = orig_fwd
    assert len(rec["calls"]) == 36 and rec["calls"][20]["frames"] == 12
    c2 = os.path.join(OUT, "config2_pipe.pt")
    if os.path.exists(c2):                              # before the switch the run IS config 2's (same noise, prompt and draws)
        assert torch.equal(lat[:, :SW], torch.load(c2)["latents"][:, :SW]), "blocks 0-3 must reproduce config 2's latents"
        rec["prefix_equals_config2"] = True
    rec["latents_after_switch"] = lat[:, SW:].clone()
    _save("config4_pipe.pt", rec)


def gen_config3(ns):
    """BASELINE config 3's regime over a longer horizon: the REFERENCE's CausalInferencePipeline (pipeline/causal_inference.py:56-253)
    at 30 layers, 60x104, single prompt, T = 48 latent frames = 16 blocks = 80 forwards: the 12-frame window fills once and then
    rolls through twelve more blocks (three full turnovers of the non-sink window).  Same noise / prompt / re-noise stream as config 2,
    so frames 0..20 reproduce `config2_pipe.pt` (asserted).  Stored (small): per block an 8192-element sample of its latents and
    24 sampled K / V slots of layers 0 / 14 / 29 + end indices after its context pass; the last two blocks' latents whole."""
    cfg = synth.longlive_1_3b()
    fs = cfg.frame_seqlen
    T, nfb = 48, 3
    sd = synth.synth_state_dict(cfg, seed=0)
    M = build_ref_model(ns, cfg, sd, fs)
    del sd
    wr = build_ref_wrapper(ns, M)
    table = {"p0": synth.synth_prompt_embeds(cfg, seed=1)}
    args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=nfb,
                           context_noise=0, global_sink=True)
    P = ns.causal_inference.CausalInferencePipeline(args, "cpu", generator=wr, text_encoder=_fake_text_encoder(table), vae=_FakeVAE())
    noise = synth.synth_noise(cfg, T, seed=0)
    S = 12 * fs
    slots = sample_rows(S, CONFIG2_SLOTS)
    samp = torch.linspace(0, nfb * 16 * cfg.lat_h * cfg.lat_w - 1, CONFIG2_NSAMPLE).round().long()
    rec = dict(T=T, slots=slots, sample_idx=samp, layers=list(CONFIG2_LAYERS), renoise_seed=43, noise_seed=0, prompt_seed=1, blocks=[])
    n_calls = [0]
    orig_fwd = wr.forward

    def spy(*a, **k):
        t1 = time.time()
        out = orig_fwd(*a, **k)
        blk, j = divmod(n_calls[0], 5)
        n_calls[0] += 1
        if j == 4:
            kv = k["kv_cache"]
            rec["blocks"].append(dict(k=[kv[l]["k"][0, slots].clone() for l in CONFIG2_LAYERS], v=[kv[l]["v"][0, slots].clone() for l in CONFIG2_LAYERS],
                                      idx=(int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"]))))
        print(f"  call {n_calls[0] - 1} (block {blk}, {'ctx' if j == 4 else 'step %d' % j}): {time.time() - t1:.1f}s", flush=True)
        return out

    real_randn_like = torch.randn_like
    wr.forward = spy
    try:
        torch.randn_like = _HashRandn(43)
        t0 = time.time()
        _, lat = P.inference(noise, ["p0"], return_latents=True)
        print(f"config3 pipeline ({n_calls[0]} forwards): {time.time() - t0:.1f}s", flush=True)
    finally:
        torch.randn_like = real_randn_like
        wr.forward = orig_fwd
    assert n_calls[0] == 80
    c2 = os.path.join(OUT, "config2_pipe.pt")
    if os.path.exists(c2):
        assert torch.equal(lat[:, :21], torch.load(c2)["latents"]), "frames 0..20 must reproduce config 2's latents"
        rec["prefix_equals_config2"] = True
    for b in range(T // nfb):
        rec["blocks"][b]["latent_sample"] = lat[:, nfb * b: nfb * b + nfb].flatten()[samp].clone()
    rec["latents_tail"] = lat[:, T - 2 * nfb:].clone()
    _save("config3_pipe.pt", rec)


def main(argv):
    from oracle import fast_hash
    fast_hash.install()                                # the same integers, ~100x faster on the CPU (host build of csrc/synth_hash.h)
    ns = load_pipelines()
    todo = argv or ["ops", "toy", "pipe", "pipe_calls", "real_block", "real_fwd"]
    for t in todo:
        t0 = time.time()
        globals()["gen_" + t](ns)
        print(f"[{t}] {time.time() - t0:.1f}s")




# --------------------------------------------------------------------------------------------
class FakeGenerator(nn.Module):
    """Records every generator call a pipeline makes and returns a cheap deterministic x0 (no model).
    Shared by the golden generator (driving the REFERENCE pipelines) and tests/test_pipeline_host.py (driving
    longlive_amd's pipelines): the two call logs must be identical."""

    def __init__(self, scheduler, frame_seq_length):
        super().__init__()
        self.dummy = nn.Parameter(torch.zeros(1))
        self.scheduler = scheduler
        self.log = []
        self.model = SimpleNamespace(num_frame_per_block=1, local_attn_size=-1, max_attention_size=0, block_mask=None,
                                     named_modules=lambda: [], _prepare_blockwise_causal_attn_mask=lambda **kw: None)
        self.fs = frame_seq_length

    def get_scheduler(self):
        return self.scheduler

    def forward(self, noisy_image_or_video, conditional_dict, timestep, kv_cache=None, crossattn_cache=None,
                current_start=None, sink_recache_after_switch=False, **kw):
        x = noisy_image_or_video
        self.log.append(dict(t=[round(float(v), 4) for v in timestep.flatten().tolist()], cs=int(current_start),
                             recache=bool(sink_recache_after_switch), prompt=conditional_dict["name"],
                             frames=int(x.shape[1]), xsum=round(float(x.float().sum()), 3),
                             kv_zero=bool(kv_cache[0]["k"].abs().sum() == 0),
                             ca_init=bool(crossattn_cache[0]["is_init"]), n_layers=len(kv_cache),
                             kv_shape=list(kv_cache[0]["k"].shape)))
        # mimic the model's side effects on the caches that the pipelines depend on
        cs, n = int(current_start), x.shape[1] * self.fs
        kv_cache[0]["k"][:, : min(n, kv_cache[0]["k"].shape[1])] += 1
        crossattn_cache[0]["is_init"] = True
        x0 = (0.5 * x.float() + 0.01 * (len(self.log) % 7)).to(x.dtype)
        return x0, x0


def gen_pipe_calls(ns):
    sch = ns.scheduler.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(1000, training=True)
    cfg = synth.WanConfig(lat_h=4, lat_w=4)          # 4 tokens per frame; heads stay 12 x 128 (hard-coded upstream)
    args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True,
                           num_frame_per_block=3, context_noise=0, global_sink=False)
    enc = lambda text_prompts: {"prompt_embeds": torch.zeros(1, 1), "name": text_prompts[0]}
    rec = {}
    real = torch.randn_like
    try:
        for tag, gs, T, switches in (("single", True, 21, None), ("inter_gs0", False, 48, [10, 16, 40]),
                                     ("inter_gs1", True, 30, [0, 20])):
            args.global_sink = gs
            fg = FakeGenerator(sch, 4)
            noise = synth.synth_noise(cfg, T, seed=3)
            torch.randn_like = _HashRandn(5)
            if switches is None:
                P = ns.causal_inference.CausalInferencePipeline(args, "cpu", generator=fg, text_encoder=enc, vae=_FakeVAE())
                P.num_transformer_blocks, P.frame_seq_length = 2, 4
                _, lat = P.inference(noise, ["p0"], return_latents=True)
            else:
                P = ns.interactive.InteractiveCausalInferencePipeline(args, "cpu", generator=fg, text_encoder=enc, vae=_FakeVAE())
                P.num_transformer_blocks, P.frame_seq_length = 2, 4
                prompts = [[f"p{i}"] for i in range(len(switches) + 1)]
                _, lat = P.inference(noise, text_prompts_list=prompts, switch_frame_indices=switches, return_latents=True)
            rec[tag] = dict(log=fg.log, latents=lat.clone(), T=T, switches=switches, global_sink=gs)
    finally:
        torch.randn_like = real
    _save("pipe_calls.pt", rec)


# --------------------------------------------------------------------------------------------
TRAIN_CASES = (  # tag, class, chunks [(frames, start, kwargs)], ctor kwargs
    ("plain", "train", dict(last_step_only=False, context_noise=0), [(6, 0, {}), (9, 6, {})]),
    ("last_step_ctx", "train", dict(last_step_only=True, context_noise=100), [(6, 0, {})]),
    ("switch_mid", "switch", dict(last_step_only=False, context_noise=0), [(6, 0, {}), (12, 6, dict(switch=6))]),
    ("switch_ext", "switch", dict(last_step_only=False, context_noise=50), [(9, 30, dict(switch=3, ext=24))]),
    ("switch_none", "switch", dict(last_step_only=True, context_noise=0), [(6, 0, dict(switch=None))]),
    ("switch_at0", "switch", dict(last_step_only=False, context_noise=0), [(6, 0, dict(switch=0))]),
)


def run_train_case(classes, fg, kind, ctor, chunks, cfg, randn_patch):
    """Drives one TRAIN_CASES entry; shared with tests/test_pipeline_host.py (which passes longlive_amd's classes)."""
    cls = classes[kind]
    P = cls(denoising_step_list=[1000, 750, 500, 250], scheduler=fg.scheduler, generator=fg, num_frame_per_block=3,
            same_step_across_blocks=False, local_attn_size=12, slice_last_frames=21, **ctor)
    P.num_transformer_blocks, P.frame_seq_length = 2, 4
    P.kv_cache_size = (12 + 21) * 4
    P._initialize_kv_cache(1, torch.bfloat16, "cpu")
    P._initialize_crossattn_cache(1, torch.bfloat16, "cpu")
    randn_patch(P)
    torch.manual_seed(11)                                   # exit steps: torch.randint on the CPU generator
    enc = lambda name: {"prompt_embeds": torch.zeros(1, 1), "name": name}
    outs, infos = [], []
    for ci, (frames, start, kw) in enumerate(chunks):
        noise = synth.synth_noise(cfg, frames, seed=3 + ci)
        args = dict(noise=noise, conditional_dict=enc("p0"), current_start_frame=start, requires_grad=False)
        if "switch" in kw:
            args.update(switch_frame_index=kw["switch"], switch_conditional_dict=enc("p1") if kw["switch"] is not None else None)
            if "ext" in kw:
                args["switch_recache_frames"] = synth.synth_noise(cfg, kw["ext"], seed=17)
        res = P.generate_chunk_with_cache(**args)
        outs.append(res[0].clone())
        infos.append(tuple(res[1:]))
    P.clear_kv_cache()
    cleared = dict(k0=float(P.kv_cache1[0]["k"].abs().sum()), g=int(P.kv_cache1[0]["global_end_index"]),
                   ca=bool(P.crossattn_cache[0]["is_init"]))
    return dict(log=fg.log, outs=outs, infos=infos, cleared=cleared)


def gen_train_calls(ns):
    """Reference StreamingTrainingPipeline / StreamingSwitchTrainingPipeline (pipeline/streaming_training.py,
    pipeline/streaming_switch_training.py) driving the fake generator with requires_grad=False."""
    import importlib
    st = importlib.import_module("pipeline.streaming_training")
    sw = importlib.import_module("pipeline.streaming_switch_training")
    classes = {"train": st.StreamingTrainingPipeline, "switch": sw.StreamingSwitchTrainingPipeline}
    sch = ns.scheduler.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(1000, training=True)
    cfg = synth.WanConfig(lat_h=4, lat_w=4)
    rec = {}
    real = torch.randn_like
    try:
        for tag, kind, ctor, chunks in TRAIN_CASES:
            fg = FakeGenerator(sch, 4)

            def patch(P):
                torch.randn_like = _HashRandn(5)
            rec[tag] = run_train_case(classes, fg, kind, ctor, chunks, cfg, patch)
            torch.randn_like = real
    finally:
        torch.randn_like = real
    _save("train_calls.pt", rec)


# --------------------------------------------------------------------------------------------
def gen_vae(ns):
    """Reference WanVAE_ decoder (wan/modules/vae.py) through the reference WanVAEWrapper.decode_to_pixel code path, with
    synthetic weights: full decode of 5 latent frames at 8x12 latents (64x96 pixels, 17 frames), and the streaming
    (use_cache) variant fed in two pieces."""
    import importlib
    vae = importlib.import_module("wan.modules.vae")
    vcfg = synth.VaeConfig()
    sd = synth.synth_vae_state_dict(vcfg, seed=5)
    model = vae.WanVAE_(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                        temperal_downsample=[False, True, True], dropout=0.0)
    missing, unexpected = model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "conv1.")) for k in missing), (missing[:4], unexpected[:4])
    model = model.to(torch.bfloat16).eval()
    W = ns.wan_wrapper.WanVAEWrapper
    wr = W.__new__(W)
    nn.Module.__init__(wr)
    wr.mean = torch.tensor(W_MEAN, dtype=torch.float32)
    wr.std = torch.tensor(W_STD, dtype=torch.float32)
    wr.model = model
    lat = synth.hash_normal(55, "vae.latent", (1, 5, 16, 8, 12)).to(torch.bfloat16)
    t0 = time.time()
    full = wr.decode_to_pixel(lat, use_cache=False)
    print(f"vae decode 5 frames @64x96: {time.time() - t0:.1f}s", tuple(full.shape))
    model.clear_cache()
    a = wr.decode_to_pixel(lat[:, :2], use_cache=True)
    b = wr.decode_to_pixel(lat[:, 2:], use_cache=True)
    model.clear_cache()
    _save("vae_decode.pt", dict(full=full.to(torch.bfloat16), full_f32_sample=full[0, :, :, ::7, ::11].clone(),
                                stream_a=a.to(torch.bfloat16), stream_b=b.to(torch.bfloat16)))


def gen_t5(ns):
    """Reference T5Encoder (wan/modules/t5.py) built by the reference's own umt5_xxl(encoder_only=True, ...) factory, cast
    to bf16 like inference.py:134, synthetic weights: (a) a tiny geometry, (b) the real widths (dim 4096, 64 heads,
    ffn 10240, L = 512) with 2 layers and a 4096-entry vocabulary.  Output = WanTextEncoder.forward's post-processing
    (padding rows zeroed, utils/wan_wrapper.py:52-53)."""
    import importlib
    t5 = importlib.import_module("wan.modules.t5")
    rec = {}
    for tag, cfg, ntok in (("tiny", synth.T5Config(vocab_size=512, dim=256, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=3,
                                                   text_len=64), 23),
                           ("wide", synth.T5Config(vocab_size=4096, num_layers=2), 77)):
        sd = synth.synth_t5_state_dict(cfg, seed=7)
        model = t5.umt5_xxl(encoder_only=True, return_tokenizer=False, dtype=torch.float32, device=torch.device("cpu"),
                            vocab_size=cfg.vocab_size, dim=cfg.dim, dim_attn=cfg.dim_attn, dim_ffn=cfg.dim_ffn,
                            num_heads=cfg.num_heads, encoder_layers=cfg.num_layers).eval().requires_grad_(False)
        model.load_state_dict({k: v.float() for k, v in sd.items()})
        model = model.to(torch.bfloat16)
        ids, mask = synth.synth_token_ids(cfg, ntok, seed=3, batch=2 if tag == "tiny" else 1)
        t0 = time.time()
        ctx = model(ids, mask)
        seq_lens = mask.gt(0).sum(dim=1).long()
        for u, v in zip(ctx, seq_lens):
            u[v:] = 0.0
        print(f"t5 {tag}: {time.time() - t0:.1f}s", tuple(ctx.shape), float(ctx.float().std()))
        rec[tag] = dict(out=ctx.clone(), ntok=ntok)
    _save("t5_enc.pt", rec)


def gen_t5_deep(ns):
    """The umT5-xxl encoder at its FULL depth: 24 layers at the real widths (dim 4096, 64 heads, ffn 10240, L = 512; a 4096-entry
    vocabulary keeps the embedding table small), the reference's own umt5_xxl(encoder_only=True) in bf16 on synthetic weights.  Only
    the final context is stored (4 MiB) plus the residual stream's per-layer RMS as a drift check.  ~5 min and ~25 GB here: the
    model is built in bf16 directly (the weights are bf16 values either way: loading them as fp32 and casting gives the same bits)."""
    import importlib
    t5 = importlib.import_module("wan.modules.t5")
    cfg = synth.T5Config(vocab_size=4096, num_layers=24)
    t0 = time.time()
    model = t5.umt5_xxl(encoder_only=True, return_tokenizer=False, dtype=torch.bfloat16, device=torch.device("cpu"),
                        vocab_size=cfg.vocab_size, dim=cfg.dim, dim_attn=cfg.dim_attn, dim_ffn=cfg.dim_ffn,
                        num_heads=cfg.num_heads, encoder_layers=cfg.num_layers).eval().requires_grad_(False)
    own = model.state_dict()
    for name, shape in synth.t5_param_shapes(cfg).items():           # one tensor at a time: never two copies of the model
        one = synth.synth_t5_state_dict(cfg, seed=7, only=name)[name]
        assert own[name].shape == one.shape and own[name].dtype == torch.bfloat16, name
        own[name].copy_(one)
    print(f"t5 deep: weights in {time.time() - t0:.1f}s")
    ids, mask = synth.synth_token_ids(cfg, 77, seed=3, batch=1)
    rms, mids = [], {}
    keep = (5, 11, 17, 23)                             # residual stream after layers 6, 12, 18, 24: the valid rows only (the padded ones never feed them)

    def hook(i):
        def f(m, inp, o):
            rms.append(float(o.float().pow(2).mean().sqrt()))
            if i in keep:
                mids[i + 1] = o[0, :77].clone()
        return f
    hooks = [blk.register_forward_hook(hook(i)) for i, blk in enumerate(model.blocks)]
    emb = model.token_embedding(ids)[0, :77].clone()   # the residual stream that enters layer 0 (dropout is the identity in eval)
    t0 = time.time()
    ctx = model(ids, mask)
    for h in hooks:
        h.remove()
    seq_lens = mask.gt(0).sum(dim=1).long()
    for u, v in zip(ctx, seq_lens):
        u[v:] = 0.0
    print(f"t5 deep: forward {time.time() - t0:.1f}s", tuple(ctx.shape), float(ctx.float().std()), [round(r, 2) for r in rms])
    _save("t5_enc_deep.pt", dict(out=ctx.clone(), ntok=77, layer_rms=rms, x_in=emb, x_after=mids))


W_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
          -0.1922, -0.9497, 0.2503, -0.2921]
W_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
         1.1253, 2.8251, 1.9160]


if __name__ == "__main__":
    main(sys.argv[1:])
