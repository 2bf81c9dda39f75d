"""Checkpoint + LoRA loading for the generator.

Mirrors what the reference's CLI does at load time (inference.py:72-130):
  * `torch.load(generator_ckpt)` holds the generator under "generator" / "generator_ema" (training checkpoints,
    trainer/distillation.py:741-816) or "model"; EMA weights carry FSDP's `_fsdp_wrapped_module.` in their names
    (inference.py:81-86);
  * the LoRA checkpoint is `{"generator_lora": sd}` or the bare dict (inference.py:113-123), as written by
    `peft.get_peft_model_state_dict` (utils/lora_utils.py:84-91): keys
    `base_model.model.<module>.lora_A.weight [r, in]` / `.lora_B.weight [out, r]` (the adapter name "default" is
    stripped by peft on save; both spellings are accepted here) for every nn.Linear inside a CausalWanAttentionBlock
    (utils/lora_utils.py:31-47): self_attn.{q,k,v,o}, cross_attn.{q,k,v,o}, ffn.0, ffn.2.

peft (unpinned in requirements.txt:40) is not vendored in the reference and not installed here; its published LoRA
forward is  y = base(x) + lora_B(lora_A(dropout(x))) * (lora_alpha / r).  The reference keeps the adapters un-merged at
inference (two extra GEMMs per linear, 300 per forward); here they are FOLDED once at load:
    W' = bf16( W + (alpha / r) * B @ A )          (fp32 arithmetic, one rounding)
so the hot path runs exactly the same kernels with zero extra work.
"""
from __future__ import annotations

import re
from typing import Dict, Mapping, Optional

import torch

_LORA_RE = re.compile(r"^(?:base_model\.model\.)?(?:model\.)?(?P<mod>.+?)\.lora_(?P<ab>[AB])(?:\.default)?\.weight$")


def extract_generator_state_dict(ckpt: Mapping, use_ema: bool = False) -> Dict[str, torch.Tensor]:
    """inference.py:72-94."""
    if "generator" in ckpt or "generator_ema" in ckpt:
        raw = ckpt["generator_ema" if use_ema else "generator"]
    elif "model" in ckpt:
        raw = ckpt["model"]
    else:
        raise ValueError("generator state dict not found (expected one of 'generator', 'generator_ema', 'model')")
    return {k.replace("_fsdp_wrapped_module.", ""): v for k, v in raw.items()}


def strip_model_prefix(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """WanDiffusionWrapper checkpoints name parameters `model.<...>`; CausalWanModelHIP's own names have no prefix."""
    return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}


def fold_lora(base_sd: Mapping[str, torch.Tensor], lora_ckpt: Mapping, rank: Optional[int] = None,
              alpha: Optional[float] = None, device=None) -> Dict[str, torch.Tensor]:
    """Returns a copy of `base_sd` (un-prefixed names) with every LoRA pair folded into its base weight.
    rank defaults to the A matrices' row count; alpha defaults to rank (utils/lora_utils.py:59-63)."""
    lora_sd = lora_ckpt["generator_lora"] if "generator_lora" in lora_ckpt else lora_ckpt
    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in lora_sd.items():
        m = _LORA_RE.match(k)
        if m is None:
            raise ValueError(f"unrecognised LoRA key: {k}")
        pairs.setdefault(m.group("mod"), {})[m.group("ab")] = v
    out = dict(base_sd)
    for mod, ab in pairs.items():
        if set(ab) != {"A", "B"}:
            raise ValueError(f"LoRA pair incomplete for {mod}: have {sorted(ab)}")
        wname = mod + ".weight"
        if wname not in out:
            raise ValueError(f"LoRA targets {mod} but the base checkpoint has no {wname}")
        A, B, W = ab["A"], ab["B"], out[wname]
        r = A.shape[0]
        if rank is not None and rank != r:
            raise ValueError(f"{mod}: LoRA rank {r} != configured rank {rank}")
        if A.shape != (r, W.shape[1]) or B.shape != (W.shape[0], r):
            raise ValueError(f"{mod}: LoRA shapes {tuple(A.shape)} / {tuple(B.shape)} do not fit weight {tuple(W.shape)}")
        scale = (alpha if alpha is not None else r) / r
        dev = device if device is not None else W.device
        folded = W.to(dev, torch.float32) + scale * (B.to(dev, torch.float32) @ A.to(dev, torch.float32))
        out[wname] = folded.to(W.dtype).to(W.device)
    return out


def load_generator(generator, generator_ckpt, lora_ckpt=None, adapter: Optional[Mapping] = None, use_ema: bool = False,
                   strict: bool = True):
    """`generator` is a longlive_amd WanDiffusionWrapper.  `generator_ckpt` / `lora_ckpt` are paths or already-loaded
    dicts; `adapter` is the yaml's adapter section (`rank`, `alpha`: configs/longlive_inference.yaml:31-37)."""
    ck = torch.load(generator_ckpt, map_location="cpu") if isinstance(generator_ckpt, (str, bytes)) else generator_ckpt
    sd = strip_model_prefix(extract_generator_state_dict(ck, use_ema))
    if lora_ckpt is not None:
        lk = torch.load(lora_ckpt, map_location="cpu") if isinstance(lora_ckpt, (str, bytes)) else lora_ckpt
        rank = adapter.get("rank") if adapter else None
        alpha = (adapter.get("alpha") or rank) if adapter else None
        dev = generator.model.patch_embedding.weight.device
        sd = fold_lora(sd, lk, rank=rank, alpha=alpha, device=dev)
    return generator.model.load_state_dict(sd, strict=strict and not use_ema)
