"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (longlive_amd/).

CPU restatement of the umT5 *encoder* as LongLive's WanTextEncoder runs it (wan/modules/t5.py:267-304 T5Encoder,
:49-64 T5LayerNorm, :66-117 T5Attention, :120-139 T5FeedForward with the python tanh-GELU of :46-50, :219-263
T5RelativeEmbedding; utils/wan_wrapper.py:16-57) in bf16 as after `pipeline.to(dtype=torch.bfloat16)` (inference.py:134).
Functional over a state dict.  Pinned bit-exact to the reference's own T5Encoder by tests/golden/t5_enc.pt
(oracle/make_golden.py::gen_t5) in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def t5_layer_norm(x: Tensor, w: Tensor, eps: float = 1e-6) -> Tensor:
    """T5LayerNorm.forward (:57-63): fp32 statistics, bf16 result, then bf16 weight multiply."""
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        y = y.type_as(w)
    return w * y


def gelu_py(x: Tensor) -> Tensor:
    """GELU.forward (:48-50): every op rounds to the tensor dtype."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def relative_buckets(lq: int, lk: int, num_buckets: int = 32, max_dist: int = 128) -> Tensor:
    """T5RelativeEmbedding._relative_position_bucket, bidirectional (:243-263): int64 [lq, lk]."""
    rel_pos = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    nb = num_buckets // 2
    rel_buckets = (rel_pos > 0).long() * nb
    rel_pos = torch.abs(rel_pos)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    rel_buckets += torch.where(rel_pos < max_exact, rel_pos, large)
    return rel_buckets


def attention(x: Tensor, sd: Dict[str, Tensor], p: str, num_heads: int, mask: Tensor, pos_bias: Tensor) -> Tensor:
    """T5Attention.forward (:85-117), self-attention: no 1/sqrt(d) scaling, additive bias, fp32 softmax."""
    b, n = x.size(0), num_heads
    c = sd[p + "q.weight"].shape[0] // n
    q = F.linear(x, sd[p + "q.weight"]).view(b, -1, n, c)
    k = F.linear(x, sd[p + "k.weight"]).view(b, -1, n, c)
    v = F.linear(x, sd[p + "v.weight"]).view(b, -1, n, c)
    attn_bias = x.new_zeros(b, n, q.size(1), k.size(1))
    attn_bias += pos_bias
    attn_bias.masked_fill_(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + attn_bias
    attn = F.softmax(attn.float(), dim=-1).type_as(attn)
    y = torch.einsum("bnij,bjnc->binc", attn, v)
    return F.linear(y.reshape(b, -1, n * c), sd[p + "o.weight"])


def encoder(ids: Tensor, mask: Tensor, sd: Dict[str, Tensor], num_layers: int, num_heads: int, num_buckets: int = 32) -> Tensor:
    """T5Encoder.forward (:296-304) with shared_pos=False: ids, mask [B, L] -> [B, L, dim]."""
    x = F.embedding(ids, sd["token_embedding.weight"])
    L = x.size(1)
    buckets = relative_buckets(L, L, num_buckets)
    for i in range(num_layers):
        p = f"blocks.{i}."
        e = F.embedding(buckets, sd[p + "pos_embedding.embedding.weight"]).permute(2, 0, 1).unsqueeze(0).contiguous()
        x = x + attention(t5_layer_norm(x, sd[p + "norm1.weight"]), sd, p + "attn.", num_heads, mask, e)
        h = t5_layer_norm(x, sd[p + "norm2.weight"])
        h = F.linear(h, sd[p + "ffn.fc1.weight"]) * gelu_py(F.linear(h, sd[p + "ffn.gate.0.weight"]))
        x = x + F.linear(h, sd[p + "ffn.fc2.weight"])
    return t5_layer_norm(x, sd["norm.weight"])


def text_encoder_forward(ids: Tensor, mask: Tensor, sd, num_layers: int, num_heads: int) -> Tensor:
    """WanTextEncoder.forward after tokenisation (utils/wan_wrapper.py:43-57): padding rows zeroed."""
    ctx = encoder(ids, mask, sd, num_layers, num_heads)
    seq_lens = mask.gt(0).sum(dim=1).long()
    for u, v in zip(ctx, seq_lens):
        u[v:] = 0.0
    return ctx
