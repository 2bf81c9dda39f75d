"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (longlive_amd/).

CPU restatement of CausalWanModel._forward_inference, WanDiffusionWrapper.forward and the two
inference pipelines, functional over a plain state dict (names as in the reference's module tree,
without the wrapper's `model.` prefix).  See oracle/ref_ops.py for the pinning story.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ref_ops as R

Tensor = torch.Tensor


@dataclass
class RefConfig:
    dim: int
    ffn_dim: int
    num_heads: int
    num_layers: int
    in_dim: int = 16
    out_dim: int = 16
    freq_dim: int = 256
    text_dim: int = 4096
    text_len: int = 512
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    eps: float = 1e-6
    local_attn_size: int = -1
    sink_size: int = 0

    @staticmethod
    def from_cfg(cfg) -> "RefConfig":
        keys = RefConfig.__dataclass_fields__.keys()
        return RefConfig(**{k: getattr(cfg, k) for k in keys})


def new_kv_cache(batch: int, cache_size: int, num_layers: int, num_heads: int, head_dim: int,
                 dtype=torch.bfloat16, device="cpu") -> List[dict]:
    """pipeline/causal_inference.py:255-279 (end indices kept as python ints: no device sync)."""
    return [dict(k=torch.zeros(batch, cache_size, num_heads, head_dim, dtype=dtype, device=device),
                 v=torch.zeros(batch, cache_size, num_heads, head_dim, dtype=dtype, device=device),
                 global_end_index=0, local_end_index=0) for _ in range(num_layers)]


def new_crossattn_cache(batch: int, text_len: int, num_layers: int, num_heads: int, head_dim: int,
                        dtype=torch.bfloat16, device="cpu") -> List[dict]:
    """pipeline/causal_inference.py:281-293."""
    return [dict(k=torch.zeros(batch, text_len, num_heads, head_dim, dtype=dtype, device=device),
                 v=torch.zeros(batch, text_len, num_heads, head_dim, dtype=dtype, device=device),
                 is_init=False) for _ in range(num_layers)]


class RefModel:
    """Functional CausalWanModel (wan/modules/causal_model.py:511-1068), KV-cache branch only."""

    def __init__(self, cfg: RefConfig, sd: Dict[str, Tensor], dtype=torch.bfloat16,
                 frame_seqlen_for_max_attn: int = 1560, lora: Optional[Dict[str, Tuple[Tensor, Tensor]]] = None,
                 lora_scaling: float = 1.0, quant: Optional[str] = None):
        self.cfg = cfg
        self.dtype = dtype
        # quant = "int8": BASELINE config 5's arithmetic (the reference ships no INT8 code, reports.md:24,39 -- this restates the
        # DEFINITION longlive_amd uses, include/longlive_hip.h ll_quantize_rows / ll_gemm_w8a8): the six per-token linears of every
        # block (self_attn q/k/v/o, cross_attn q/o, ffn.0, ffn.2) run W8A8 -- weights quantised per output channel, activations per
        # token, symmetric, scale = max|row| / 127 (1 for an all-zero row), q = clamp(rint(x * (1 / scale)), -127, 127), exact
        # integer accumulation, y = bf16(float(acc) * (sx[m] * sw[n]) + bias), every step in fp32 in that order.
        assert quant in (None, "int8")
        self.quant = quant
        self._wq: Dict[str, Tuple[Tensor, Tensor]] = {}
        # un-merged LoRA adapters {module name: (A [r,in], B [out,r])}: peft's published forward
        #   y = base(x) + lora_B(lora_A(x)) * scaling   (what the reference runs at inference, inference.py:97-130)
        self.lora = {k: (a.to(dtype), b.to(dtype)) for k, (a, b) in (lora or {}).items()}
        self.lora_scaling = lora_scaling
        self.sd = {k: v.to(dtype) for k, v in sd.items()}
        self.freqs = R.make_freqs(cfg.dim // cfg.num_heads)
        # causal_model.py:82-88 hard-codes 1560 tokens/frame in max_attention_size; the pipelines
        # overwrite it with local_attn_size * frame_seq_length (causal_inference.py:295-329).
        self.max_attention_size = (32760 if cfg.local_attn_size == -1
                                   else cfg.local_attn_size * frame_seqlen_for_max_attn)
        self.local_attn_size = cfg.local_attn_size

    # -- helpers ----------------------------------------------------------------------------
    @staticmethod
    def quantize_rows(x: Tensor) -> Tuple[Tensor, Tensor]:
        """(q float64 holding the int8 values, scale fp32 [rows]) of a [rows, K] tensor."""
        xf = x.float()
        mx = xf.abs().amax(dim=-1)
        sc = torch.where(mx > 0, mx / 127.0, torch.ones_like(mx))
        inv = (1.0 / sc).unsqueeze(-1)
        return torch.round(xf * inv).clamp_(-127, 127).double(), sc

    _W8A8 = (".self_attn.q", ".self_attn.k", ".self_attn.v", ".self_attn.o", ".cross_attn.q", ".cross_attn.o", ".ffn.0", ".ffn.2")

    def lin(self, x: Tensor, name: str) -> Tensor:
        if self.quant == "int8" and name.startswith("blocks.") and name.endswith(self._W8A8):
            assert name not in self.lora
            if name not in self._wq:
                self._wq[name] = self.quantize_rows(self.sd[name + ".weight"])
            wq, sw = self._wq[name]
            xq, sx = self.quantize_rows(x.reshape(-1, x.shape[-1]).to(self.dtype))
            acc = (xq @ wq.t()).float()                                  # exact: |acc| < 2^31 << 2^53; then int32 -> fp32 (RN)
            y = acc * (sx.unsqueeze(1) * sw.unsqueeze(0)) + self.sd[name + ".bias"].float()
            return y.to(self.dtype).reshape(*x.shape[:-1], -1)
        y = F.linear(x, self.sd[name + ".weight"], self.sd.get(name + ".bias"))
        if name in self.lora:
            a, b = self.lora[name]
            y = y + F.linear(F.linear(x, a), b) * self.lora_scaling
        return y

    # -- self attention (causal_model.py:97-370, KV branch :205-370) --------------------------
    def self_attn(self, x: Tensor, p: str, grid, kv_cache: dict, current_start: int,
                  sink_recache_after_switch: bool) -> Tuple[Tensor, dict]:
        c = self.cfg
        b, s, n, d = x.shape[0], x.shape[1], c.num_heads, c.dim // c.num_heads
        q = R.rms_norm(self.lin(x, p + "q"), self.sd[p + "norm_q.weight"], c.eps).view(b, s, n, d)
        k = R.rms_norm(self.lin(x, p + "k"), self.sd[p + "norm_k.weight"], c.eps).view(b, s, n, d)
        v = self.lin(x, p + "v").view(b, s, n, d)
        frame_seqlen = grid[1] * grid[2]
        start_frame = current_start // frame_seqlen
        rq = R.causal_rope_apply(q, grid, self.freqs, start_frame).type_as(v)
        rk = R.causal_rope_apply(k, grid, self.freqs, start_frame).type_as(v)
        plan = R.kv_plan(current_start, s, kv_cache["global_end_index"], kv_cache["local_end_index"],
                         kv_cache["k"].shape[1], c.sink_size * frame_seqlen, self.local_attn_size,
                         self.max_attention_size, sink_recache_after_switch)
        # the reference works on a clone and commits after the last layer (:849-905); each layer owns
        # its cache, so the in-place commit here is equivalent -- but the END INDICES must only move
        # after all layers have planned with the old values, hence they are committed by the caller.
        R.kv_apply(kv_cache["k"], kv_cache["v"], plan, rk, v)
        kc, vc = R.kv_gather(kv_cache["k"], kv_cache["v"], plan)
        y = R.attention(rq, kc, vc, dtype=self.dtype)
        return self.lin(y.flatten(2), p + "o"), plan

    # -- cross attention (model.py:159-194) ---------------------------------------------------
    def cross_attn(self, x: Tensor, p: str, context: Tensor, cache: Optional[dict]) -> Tensor:
        c = self.cfg
        b, n, d = x.shape[0], c.num_heads, c.dim // c.num_heads
        q = R.rms_norm(self.lin(x, p + "q"), self.sd[p + "norm_q.weight"], c.eps).view(b, -1, n, d)
        if cache is not None and cache["is_init"]:
            k, v = cache["k"], cache["v"]
        else:
            k = R.rms_norm(self.lin(context, p + "k"), self.sd[p + "norm_k.weight"], c.eps).view(b, -1, n, d)
            v = self.lin(context, p + "v").view(b, -1, n, d)
            if cache is not None:
                cache["is_init"] = True
                cache["k"], cache["v"] = k, v
        y = R.attention(q, k, v, dtype=self.dtype)   # model.py:189, all text_len positions (k_lens=None)
        return self.lin(y.flatten(2), p + "o")

    # -- block (causal_model.py:413-477) --------------------------------------------------------
    def block(self, x: Tensor, i: int, e0: Tensor, grid, context: Tensor, kv_cache: dict,
              crossattn_cache: Optional[dict], current_start: int, sink_recache_after_switch: bool):
        c = self.cfg
        p = f"blocks.{i}."
        nf = e0.shape[1]
        fs = x.shape[1] // nf
        e = (self.sd[p + "modulation"].unsqueeze(1) + e0).chunk(6, dim=2)            # :440
        y, plan = self.self_attn(R.ln_modulate(x, e[1], e[0], nf, c.eps), p + "self_attn.", grid,
                                 kv_cache, current_start, sink_recache_after_switch)   # :444-447
        x = x + (y.unflatten(1, (nf, fs)) * e[2]).flatten(1, 2)                       # :456
        xn = R.layer_norm(x, c.eps, self.sd[p + "norm3.weight"], self.sd[p + "norm3.bias"])
        x = x + self.cross_attn(xn, p + "cross_attn.", context, crossattn_cache)      # :460
        h = self.lin(R.ln_modulate(x, e[4], e[3], nf, c.eps), p + "ffn.0")
        y = self.lin(F.gelu(h, approximate="tanh"), p + "ffn.2")                      # :462-465
        x = x + (y.unflatten(1, (nf, fs)) * e[5]).flatten(1, 2)                       # :467
        return x, plan

    # -- embeddings -------------------------------------------------------------------------
    def time_embed(self, t: Tensor) -> Tuple[Tensor, Tensor]:
        """causal_model.py:976-979: e [B*F, C], e0 [B, F, 6, C]."""
        c = self.cfg
        emb = R.sinusoidal_embedding_1d(c.freq_dim, t.flatten()).to(self.dtype)
        e = self.lin(F.silu(self.lin(emb, "time_embedding.0")), "time_embedding.2")
        e0 = self.lin(F.silu(e), "time_projection.1").unflatten(1, (6, c.dim)).unflatten(0, t.shape)
        return e, e0

    def text_embed(self, context: Tensor) -> Tensor:
        """causal_model.py:984-989 (context already padded to text_len)."""
        h = F.gelu(self.lin(context.to(self.dtype), "text_embedding.0"), approximate="tanh")
        return self.lin(h, "text_embedding.2")

    # -- full forward (causal_model.py:907-1068) ------------------------------------------------
    def forward(self, x: Tensor, t: Tensor, context: Tensor, kv_cache: List[dict],
                crossattn_cache: List[dict], current_start: int = 0,
                sink_recache_after_switch: bool = False) -> Tensor:
        """x [B, C, F, H, W]; t [B, F]; context [B, text_len, text_dim] -> flow [B, C, F, H, W]."""
        c = self.cfg
        B, _, nf, H, W = x.shape
        w = self.sd["patch_embedding.weight"]
        xe = F.conv3d(x.to(self.dtype), w, self.sd["patch_embedding.bias"], stride=c.patch_size)  # :959
        grid = tuple(xe.shape[2:])
        xe = xe.flatten(2).transpose(1, 2)                                              # :963
        e, e0 = self.time_embed(t)
        ctx = self.text_embed(context)
        plans = []
        for i in range(c.num_layers):
            xe, plan = self.block(xe, i, e0, grid, ctx, kv_cache[i],
                                  crossattn_cache[i] if crossattn_cache is not None else None,
                                  current_start, sink_recache_after_switch)
            plans.append(plan)
        for i, plan in enumerate(plans):                                                # :901-904
            kv_cache[i]["global_end_index"] = plan["G_new"]
            kv_cache[i]["local_end_index"] = plan["E_new"]
        # head (:497-508) with e (not e0), 2-way modulation
        eh = (self.sd["head.modulation"].unsqueeze(1) + e.unflatten(0, t.shape).unsqueeze(2)).chunk(2, dim=2)
        fs = xe.shape[1] // nf
        y = self.lin(R.layer_norm(xe, c.eps).unflatten(1, (nf, fs)) * (1 + eh[1]) + eh[0], "head.head")
        # unpatchify (:1240-1263)
        y = y.reshape(B, *grid, *c.patch_size, c.out_dim)
        y = torch.einsum("bfhwpqrc->bcfphqwr", y)
        return y.reshape(B, c.out_dim, *[g * p for g, p in zip(grid, c.patch_size)])


class RefGenerator:
    """WanDiffusionWrapper.forward (utils/wan_wrapper.py:224-300) around RefModel."""

    def __init__(self, model: RefModel, timestep_shift: float = 5.0):
        self.model = model
        self.scheduler = R.FlowMatchSchedulerRef(shift=timestep_shift)

    def __call__(self, noisy: Tensor, prompt_embeds: Tensor, timestep: Tensor, kv_cache, crossattn_cache,
                 current_start: int, sink_recache_after_switch: bool = False):
        flow = self.model.forward(noisy.permute(0, 2, 1, 3, 4), timestep, prompt_embeds, kv_cache,
                                  crossattn_cache, current_start, sink_recache_after_switch
                                  ).permute(0, 2, 1, 3, 4)
        x0 = R.flow_to_x0(self.scheduler, flow.flatten(0, 1), noisy.flatten(0, 1),
                          timestep.flatten(0, 1)).unflatten(0, flow.shape[:2])
        return flow, x0


def run_pipeline(gen: RefGenerator, noise: Tensor, prompt_embeds_list: Sequence[Tensor],
                 switch_frame_indices: Sequence[int], denoising_step_list: Sequence[int],
                 num_frame_per_block: int, frame_seqlen: int, context_noise: int = 0,
                 global_sink: bool = True, renoise: Optional[Callable[[int, int, Tensor], Tensor]] = None,
                 on_block: Optional[Callable[[int], None]] = None) -> Tensor:
    """Latent-level restatement of CausalInferencePipeline.inference (pipeline/causal_inference.py:56-253)
    and, with >1 prompt, of InteractiveCausalInferencePipeline.inference + _recache_after_switch
    (pipeline/interactive_causal_inference.py:34-106,108-432).  Text encoder and VAE are outside the hot
    path: prompts arrive as embeddings, latents are returned.

    `renoise(block, step, like)` supplies the noise the reference draws with torch.randn_like
    (causal_inference.py:175) so that multi-step runs are reproducible across devices."""
    m = gen.model
    c = m.cfg
    B, T = noise.shape[:2]
    assert T % num_frame_per_block == 0
    steps = R.warp_denoising_steps(gen.scheduler, list(denoising_step_list))
    hd = c.dim // c.num_heads
    cache_size = (c.local_attn_size if c.local_attn_size != -1 else T) * frame_seqlen
    kv = new_kv_cache(B, cache_size, c.num_layers, c.num_heads, hd, noise.dtype)
    ca = new_crossattn_cache(B, c.text_len, c.num_layers, c.num_heads, hd, noise.dtype)
    m.max_attention_size = 32760 if c.local_attn_size == -1 else c.local_attn_size * frame_seqlen
    output = torch.zeros_like(noise)
    seg = 0
    start = 0
    for blk in range(T // num_frame_per_block):
        nf = num_frame_per_block
        if seg < len(switch_frame_indices) and start >= switch_frame_indices[seg]:
            seg += 1
            # _recache_after_switch (interactive_causal_inference.py:34-106)
            if not global_sink:
                for cch in kv:
                    cch["k"].zero_(); cch["v"].zero_()
            for cch in ca:
                cch["k"] = torch.zeros_like(cch["k"]); cch["v"] = torch.zeros_like(cch["v"]); cch["is_init"] = False
            if start > 0:
                nre = start if c.local_attn_size == -1 else min(c.local_attn_size, start)
                frames = output[:, start - nre:start]
                ts = torch.ones([B, nre], dtype=torch.int64) * context_noise
                gen(frames, prompt_embeds_list[seg], ts, kv, ca, (start - nre) * frame_seqlen,
                    sink_recache_after_switch=not global_sink)
                for cch in ca:
                    cch["k"] = torch.zeros_like(cch["k"]); cch["v"] = torch.zeros_like(cch["v"]); cch["is_init"] = False
        cond = prompt_embeds_list[seg]
        noisy = noise[:, start:start + nf]
        for idx, cur_t in enumerate(steps):
            timestep = torch.ones([B, nf], dtype=torch.int64) * cur_t
            _, x0 = gen(noisy, cond, timestep, kv, ca, start * frame_seqlen)
            if idx < len(steps) - 1:
                nxt = steps[idx + 1]
                eps_ = (renoise(blk, idx, x0.flatten(0, 1)) if renoise is not None
                        else torch.randn_like(x0.flatten(0, 1)))
                noisy = gen.scheduler.add_noise(
                    x0.flatten(0, 1), eps_, nxt * torch.ones([B * nf], dtype=torch.long)
                ).unflatten(0, x0.shape[:2])
        output[:, start:start + nf] = x0
        ctx_t = torch.ones_like(timestep) * context_noise
        gen(x0, cond, ctx_t, kv, ca, start * frame_seqlen)
        if on_block is not None:
            on_block(blk)
        start += nf
    return output
