"""umT5-xxl text encoder on the MI355X: the `text_encoder` object the reference pipelines
call once per prompt (`self.text_encoder(text_prompts=...)`, pipeline/causal_inference.py:88-90 ->
utils/wan_wrapper.py:16-57 WanTextEncoder), with the reference's state-dict names (wan/modules/t5.py::T5Encoder) and its
bf16 arithmetic (inference.py:134 casts the pipeline to bfloat16).

24 x [T5 RMS norm -> fused QK GEMM + V^T GEMM -> bias-added un-scaled attention -> O GEMM (+residual) -> norm -> fused
gate|fc1 GEMM -> python-GELU gate -> fc2 GEMM (+residual)] on the 512 padded text positions: 4.9 TFLOP per prompt.
V is produced already transposed (Wv x^T: the same GEMM with operands swapped, T5 linears have no bias), which is the
layout the attention kernel's second contraction reads.  The relative-position bias (t5.py:219-263) is a [heads, 2L-1]
table per layer built once at load.  Tokenisation (sentencepiece via HuggingfaceTokenizer, tokenizers.py) stays on the
host and is injected: the checkpoint's tokenizer files are assets, not code.  There is no CPU path.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .synth import T5Config, t5_param_shapes

bf16 = torch.bfloat16


def relative_bucket_table(L: int, num_buckets: int = 32, max_dist: int = 128) -> torch.Tensor:
    """bucket(rel) for rel = -(L-1) .. L-1 (key index minus query index), bidirectional T5 bucketing
    (T5RelativeEmbedding._relative_position_bucket, t5.py:243-263): int64 [2L-1]."""
    rel = torch.arange(-(L - 1), L)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(n < max_exact, n, large)


class UMT5EncoderHIP(nn.Module):
    """T5Encoder (wan/modules/t5.py:267-304), shared_pos=False, inference only."""

    def __init__(self, cfg: Optional[T5Config] = None, device="cuda"):
        super().__init__()
        self.cfg = cfg or T5Config()
        c = self.cfg
        assert c.dim_attn == c.num_heads * 64, "the attention kernel is built for head_dim 64 (umT5-xxl)"
        self._names: Dict[str, str] = {}
        for name, shape in t5_param_shapes(c).items():
            reg = name.replace(".", "__")
            self._names[name] = reg
            self.register_parameter(reg, nn.Parameter(torch.empty(shape, dtype=bf16, device=device), requires_grad=False))
        self._packed = False

    def state_dict(self, *a, **k):
        return {name: getattr(self, reg).data for name, reg in self._names.items()}

    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        missing = [n for n in self._names if n not in sd]
        unexpected = [n for n in sd if n not in self._names]
        if strict and (missing or unexpected):
            raise RuntimeError(f"UMT5EncoderHIP.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for name, reg in self._names.items():
            if name in sd:
                p = getattr(self, reg)
                if tuple(sd[name].shape) != tuple(p.shape):
                    raise RuntimeError(f"{name}: shape {tuple(sd[name].shape)} != {tuple(p.shape)}")
                p.data.copy_(sd[name].to(device=p.device, dtype=bf16))
        self._packed = False
        return missing, unexpected

    def _p(self, name):
        return getattr(self, self._names[name]).data

    def _pack(self):
        c, dev = self.cfg, self._p("norm.weight").device
        L = c.text_len
        buckets = relative_bucket_table(L, c.num_buckets, c.max_dist).to(dev)
        self._layers = []
        for i in range(c.num_layers):
            p = f"blocks.{i}."
            self._layers.append(dict(
                n1=self._p(p + "norm1.weight"), n2=self._p(p + "norm2.weight"),
                wqk=torch.cat([self._p(p + "attn.q.weight"), self._p(p + "attn.k.weight")], 0).contiguous(),
                wv=self._p(p + "attn.v.weight"), wo=self._p(p + "attn.o.weight"),
                wgf=torch.cat([self._p(p + "ffn.gate.0.weight"), self._p(p + "ffn.fc1.weight")], 0).contiguous(),
                w2=self._p(p + "ffn.fc2.weight"),
                bias=self._p(p + "pos_embedding.embedding.weight")[buckets].t().contiguous()))     # [H, 2L-1]
        z = lambda n: torch.zeros(n, dtype=bf16, device=dev)
        self._zeros = {n: z(n) for n in {2 * c.dim_attn, L, c.dim, 2 * c.dim_ffn}}
        self._packed = True

    @torch.no_grad()
    def forward(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """ids, mask int64 [B, text_len] (HuggingfaceTokenizer output, padding='max_length') -> context [B, L, dim] bf16."""
        c = self.cfg
        if not self._packed:
            self._pack()
        B, L = ids.shape
        if L != c.text_len:
            raise RuntimeError(f"UMT5EncoderHIP: expected {c.text_len} padded positions, got {L}")
        ids_h, mask_h = ids.cpu(), mask.cpu()
        if int(ids_h.min()) < 0 or int(ids_h.max()) >= c.vocab_size:
            raise RuntimeError("UMT5EncoderHIP: token id outside the vocabulary")
        dev = self._p("norm.weight").device
        ids_d = ids_h.to(dev)
        zq = self._zeros
        outs = []
        for b in range(B):
            m = mask_h[b]
            n = int(m.gt(0).sum())
            if n < 1 or not bool((m[:n] > 0).all()):
                raise RuntimeError("UMT5EncoderHIP: mask must be a non-empty prefix (right-padded prompts)")
            x = ops.gather_rows(self._p("token_embedding.weight"), ids_d[b].contiguous())
            _, h = self.run_layers(x, n, 0, len(self._layers))
            outs.append(h)
        return torch.stack(outs, 0)

    def run_layers(self, x: torch.Tensor, n_valid: int, first: int, last: int):
        """Layers [first, last) on the residual stream x [L, dim] of ONE prompt with n_valid unpadded positions -> (x after layer
        last - 1, the norm that follows it applied: layer `last`'s norm1, or the encoder's final norm after the last layer).
        forward() is run_layers(embedding, n, 0, num_layers); the tests enter mid-stack with the reference's own hidden states."""
        c = self.cfg
        if not self._packed:
            self._pack()
        L = x.shape[0]
        zq = self._zeros
        h = ops.t5_rmsnorm(x, self._layers[first]["n1"])
        for i in range(first, last):
            ly = self._layers[i]
            qk = ops.gemm(h, ly["wqk"], zq[2 * c.dim_attn])
            vt = ops.gemm(ly["wv"], h, zq[L])                                    # [dim_attn, L] = Wv h^T
            a = ops.t5_attention(qk, vt, ly["bias"], c.num_heads, n_valid)
            # x = x + o(a); h = norm2(x)   and   x = x + w2(g); h = the next layer's norm1(x) (the encoder's final norm after
            # the last layer): the residual update and the norm that follows it are one call (same bits as the two kernels)
            x, h = ops.gemm_res_t5norm(a, ly["wo"], zq[c.dim], x, ly["n2"])
            g = ops.t5_gated_gelu(ops.gemm(h, ly["wgf"], zq[2 * c.dim_ffn]))
            nxt = self._layers[i + 1]["n1"] if i + 1 < len(self._layers) else self._p("norm.weight")
            x, h = ops.gemm_res_t5norm(g, ly["w2"], zq[c.dim], x, nxt)
        return x, h


class WanTextEncoder(nn.Module):
    """Drop-in for utils/wan_wrapper.py::WanTextEncoder: `forward(text_prompts) -> {"prompt_embeds": [B, 512, 4096]}`
    with padding rows zeroed.  `tokenizer(texts) -> (ids, mask)` is injected (the reference's
    HuggingfaceTokenizer(name=..., seq_len=512, clean='whitespace') called with return_mask=True,
    add_special_tokens=True); `encode_ids` is the device path proper."""

    def __init__(self, cfg: Optional[T5Config] = None, device="cuda", tokenizer: Optional[Callable] = None):
        super().__init__()
        self.text_encoder = UMT5EncoderHIP(cfg, device=device)
        self.tokenizer = tokenizer

    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        sd = {(k[len("text_encoder."):] if k.startswith("text_encoder.") else k): v for k, v in sd.items()}
        return self.text_encoder.load_state_dict(sd, strict=strict)

    @torch.no_grad()
    def encode_ids(self, ids: torch.Tensor, mask: torch.Tensor) -> dict:
        ctx = self.text_encoder(ids, mask)
        for u, v in zip(ctx, mask.gt(0).sum(dim=1).tolist()):
            u[v:] = 0.0                                                              # utils/wan_wrapper.py:52-53
        return {"prompt_embeds": ctx}

    def forward(self, text_prompts: List[str]) -> dict:
        if self.tokenizer is None:
            raise RuntimeError("WanTextEncoder: no tokenizer injected (pass tokenizer=HuggingfaceTokenizer(...) "
                               "or call encode_ids(ids, mask))")
        try:
            ids, mask = self.tokenizer(text_prompts, return_mask=True, add_special_tokens=True)
        except TypeError:
            ids, mask = self.tokenizer(text_prompts)
        return self.encode_ids(ids, mask)
