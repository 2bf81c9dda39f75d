"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (longlive_amd/).

CPU restatement (plain PyTorch, no reference imports) of the arithmetic on
LongLive's frame-level autoregressive inference hot path.  Every function cites
the reference file:line it follows (paths relative to the reference root).

Parity pin: the reference holds no tests for this path, so the
pin is `tests/golden/*.pt`: outputs of the *reference's own modules* run on CPU
in the build container by `oracle/make_golden.py` (which imports them through
`oracle/ref_import.py`).  `tests/test_oracle_golden.py` checks this restatement
against those vectors bit-for-bit (same torch ops in the same order).

`dtype=torch.bfloat16` reproduces the reference's rounding points (the model is run
as `pipeline.to(dtype=torch.bfloat16)`, inference.py:134).  `dtype=torch.float32`
is the "truth" mode used to state the floating-point tolerance: same bf16-rounded
weights and inputs, all arithmetic in fp32.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# embeddings / tables
# ----------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim: int, position: Tensor) -> Tensor:
    """wan/modules/model.py:15-25 (float64)."""
    half = dim // 2
    position = position.type(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_params(max_seq_len: int, dim: int, theta: float = 10000) -> Tensor:
    """wan/modules/model.py:29-36 (complex128 table)."""
    freqs = torch.outer(
        torch.arange(max_seq_len),
        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def make_freqs(head_dim: int) -> Tensor:
    """wan/modules/causal_model.py:622-629: [1024, head_dim/2] complex128 (22|21|21 split at d=128)."""
    d = head_dim
    return torch.cat([
        rope_params(1024, d - 4 * (d // 6)),
        rope_params(1024, 2 * (d // 6)),
        rope_params(1024, 2 * (d // 6)),
    ], dim=1)


def causal_rope_apply(x: Tensor, grid: Tuple[int, int, int], freqs: Tensor, start_frame: int = 0) -> Tensor:
    """wan/modules/causal_model.py:32-60.  x [B, L, n, d]; grid = (F, H, W) identical for the batch."""
    n, c = x.size(2), x.size(3) // 2
    fr = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    f, h, w = grid
    seq_len = f * h * w
    out = []
    for i in range(x.size(0)):
        x_i = torch.view_as_complex(x[i, :seq_len].to(torch.float64).reshape(seq_len, n, -1, 2))
        freqs_i = torch.cat([
            fr[0][start_frame:start_frame + f].view(f, 1, 1, -1).expand(f, h, w, -1),
            fr[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
            fr[2][:w].view(1, 1, w, -1).expand(f, h, w, -1),
        ], dim=-1).reshape(seq_len, 1, -1)
        x_i = torch.view_as_real(x_i * freqs_i).flatten(2)
        x_i = torch.cat([x_i, x[i, seq_len:]])
        out.append(x_i)
    return torch.stack(out).type_as(x)


# ----------------------------------------------------------------------------
# scheduler (utils/scheduler.py:106-176) and flow <-> x0 (utils/wan_wrapper.py:175-199)
# ----------------------------------------------------------------------------
class FlowMatchSchedulerRef:
    """utils/scheduler.py:108-141 with the arguments of utils/wan_wrapper.py:141-144
    (shift=timestep_shift, sigma_min=0.0, extra_one_step=True; set_timesteps(1000, training=True))."""

    def __init__(self, shift: float = 5.0, sigma_min: float = 0.0, sigma_max: float = 1.0,
                 num_train_timesteps: int = 1000, num_inference_steps: int = 1000):
        sigmas = torch.linspace(sigma_min + (sigma_max - sigma_min) * 1.0, sigma_min,
                                num_inference_steps + 1)[:-1]
        self.sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.timesteps = self.sigmas * num_train_timesteps

    def add_noise(self, original_samples: Tensor, noise: Tensor, timestep: Tensor) -> Tensor:
        """utils/scheduler.py:159-176."""
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        timestep_id = torch.argmin((self.timesteps.unsqueeze(0) - timestep.unsqueeze(1)).abs(), dim=1)
        sigma = self.sigmas[timestep_id].reshape(-1, 1, 1, 1)
        sample = (1 - sigma) * original_samples + sigma * noise
        return sample.type_as(noise)


def warp_denoising_steps(sched: FlowMatchSchedulerRef, steps: List[int]) -> Tensor:
    """pipeline/causal_inference.py:33-37."""
    idx = torch.tensor(steps, dtype=torch.long)
    timesteps = torch.cat((sched.timesteps.cpu(), torch.tensor([0], dtype=torch.float32)))
    return timesteps[1000 - idx]


def flow_to_x0(sched: FlowMatchSchedulerRef, flow_pred: Tensor, xt: Tensor, timestep: Tensor) -> Tensor:
    """utils/wan_wrapper.py:175-199 (float64).  flow_pred/xt [N, C, H, W]; timestep [N]."""
    dt = flow_pred.dtype
    fp, x, sig, ts = (a.double() for a in (flow_pred, xt, sched.sigmas, sched.timesteps))
    tid = torch.argmin((ts.unsqueeze(0) - timestep.unsqueeze(1)).abs(), dim=1)
    sigma_t = sig[tid].reshape(-1, 1, 1, 1)
    return (x - sigma_t * fp).to(dt)


# ----------------------------------------------------------------------------
# norms
# ----------------------------------------------------------------------------
def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """wan/modules/model.py:78-86: fp32 normalise -> cast to x.dtype -> * weight."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return y.type_as(x) * weight


def layer_norm(x: Tensor, eps: float, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None) -> Tensor:
    """wan/modules/model.py:89-99."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps).type_as(x)


def ln_modulate(x: Tensor, scale: Tensor, shift: Tensor, num_frames: int, eps: float) -> Tensor:
    """wan/modules/causal_model.py:445: (norm(x).unflatten(1,(F,fs)) * (1 + scale) + shift).flatten(1,2);
    scale/shift [B, F, 1, C]."""
    fs = x.shape[1] // num_frames
    return (layer_norm(x, eps).unflatten(1, (num_frames, fs)) * (1 + scale) + shift).flatten(1, 2)


# ----------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------
def attention(q: Tensor, k: Tensor, v: Tensor, dtype=torch.bfloat16) -> Tensor:
    """wan/modules/attention.py:182-197 (the SDPA fallback the reference takes without flash-attn):
    q [B, Lq, n, d], k/v [B, Lk, n, d]; no mask, not causal, scale 1/sqrt(d)."""
    q = q.transpose(1, 2).to(dtype)
    k = k.transpose(1, 2).to(dtype)
    v = v.transpose(1, 2).to(dtype)
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=None, is_causal=False, dropout_p=0.)
    return out.transpose(1, 2).contiguous()


def attention_exact(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """fp64 softmax(QK^T/sqrt(d))V of the same tensors: the kernel-level reference for flash kernels."""
    qd, kd, vd = (t.transpose(1, 2).double() for t in (q, k, v))
    s = qd @ kd.transpose(-1, -2) / math.sqrt(q.shape[-1])
    return (torch.softmax(s, dim=-1) @ vd).transpose(1, 2).contiguous()


# ----------------------------------------------------------------------------
# KV-cache state machine (wan/modules/causal_model.py:205-360, 849-905)
# ----------------------------------------------------------------------------
def kv_plan(current_start: int, num_new: int, G: int, E: int, cache_size: int, sink_tokens: int,
            local_attn_size: int, max_attention_size: int, sink_recache_after_switch: bool = False) -> dict:
    """Pure-integer restatement of the index bookkeeping in CausalWanSelfAttention.forward.

    G = global_end_index, E = local_end_index (both BEFORE the call).  Returns what is rolled, what is
    written where, which slots attention reads, and the committed (G', E')."""
    current_end = current_start + num_new
    is_recompute = current_end <= G and current_start > 0                      # :230
    plan = dict(current_end=current_end, is_recompute=is_recompute, roll=None)
    if local_attn_size != -1 and current_end > G and num_new + E > cache_size:  # :231-232
        evict = num_new + E - cache_size                                        # :235
        rolled = E - evict - sink_tokens                                        # :236
        local_end = E + current_end - G - evict                                 # :244-245
        local_start = local_end - num_new                                       # :246
        plan["roll"] = dict(dst=sink_tokens, src=sink_tokens + evict, n=rolled)  # :257-260
        write_start = max(local_start, sink_tokens) if is_recompute else local_start  # :264
    else:
        local_end = E + current_end - G                                         # :293
        local_start = local_end - num_new                                       # :294
        write_start = max(local_start, sink_tokens) if is_recompute else local_start  # :302
        if sink_recache_after_switch:
            write_start = local_start                                           # :303-304
    roped_offset = max(0, write_start - local_start)                            # :265,305
    write_len = max(0, local_end - write_start)                                 # :266,306
    plan.update(local_start=local_start, local_end=local_end, write_start=write_start,
                roped_offset=roped_offset, write_len=write_len)
    if sink_tokens > 0:                                                         # :331-353
        local_budget = max_attention_size - sink_tokens
        if local_budget > 0:
            ws = max(sink_tokens, local_end - local_budget)
            plan["segments"] = [(0, sink_tokens), (ws, local_end)]
        else:
            plan["segments"] = [(0, sink_tokens)]
    else:                                                                       # :355-360
        plan["segments"] = [(max(0, local_end - max_attention_size), local_end)]
    # commit (:901-904): indices advance unless is_recompute
    plan["G_new"] = G if is_recompute else current_end
    plan["E_new"] = E if is_recompute else local_end
    return plan


def kv_apply(cache_k: Tensor, cache_v: Tensor, plan: dict, new_k: Tensor, new_v: Tensor) -> None:
    """In-place roll + insert on [B, S, n, d] caches (causal_model.py:257-269,310-311 == 874-897)."""
    r = plan["roll"]
    if r is not None and r["n"] > 0:
        cache_k[:, r["dst"]:r["dst"] + r["n"]] = cache_k[:, r["src"]:r["src"] + r["n"]].clone()
        cache_v[:, r["dst"]:r["dst"] + r["n"]] = cache_v[:, r["src"]:r["src"] + r["n"]].clone()
    if plan["write_len"] > 0:
        ws, ro, wl = plan["write_start"], plan["roped_offset"], plan["write_len"]
        cache_k[:, ws:ws + wl] = new_k[:, ro:ro + wl]
        cache_v[:, ws:ws + wl] = new_v[:, ro:ro + wl]


def kv_gather(cache_k: Tensor, cache_v: Tensor, plan: dict) -> Tuple[Tensor, Tensor]:
    ks = [cache_k[:, a:b] for a, b in plan["segments"]]
    vs = [cache_v[:, a:b] for a, b in plan["segments"]]
    if len(ks) == 1:
        return ks[0], vs[0]
    return torch.cat(ks, dim=1), torch.cat(vs, dim=1)
