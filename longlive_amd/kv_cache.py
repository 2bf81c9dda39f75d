"""Host-side KV-cache bookkeeping for the frame-sink + sliding-window self-attention.

Pure integer logic, restating CausalWanSelfAttention.forward's index arithmetic
(wan/modules/causal_model.py:205-360) and the index commit of _apply_cache_updates (:901-904).
The reference reads its end indices back from device memory with six `.item()` syncs per layer
(:230-245,293); here they are python ints kept beside the cache (`KVState`), so a forward never syncs.

Cache layout is the reference's (pipeline/causal_inference.py:271-277): per layer k, v = [B, S, H, D] bf16,
slot order = [sink | window], new tokens appended at local_end; when full, the window part is shifted left
by the evicted token count while the sink stays.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple


@dataclass
class KVPlan:
    current_end: int
    is_recompute: bool
    roll: Optional[Tuple[int, int, int]]   # (dst, src, n) rows to shift left, or None
    local_start: int
    local_end: int
    write_start: int                       # first cache slot written
    roped_offset: int                      # first new token that is written
    write_len: int
    segments: List[Tuple[int, int]]        # [start, end) slot ranges attention reads, in order
    G_new: int
    E_new: int


def plan_update(current_start: int, num_new: int, G: int, E: int, cache_size: int, sink_tokens: int,
                local_attn_size: int, max_attention_size: int, sink_recache_after_switch: bool = False) -> KVPlan:
    """G = global_end_index, E = local_end_index before the call."""
    current_end = current_start + num_new
    is_recompute = current_end <= G and current_start > 0                           # :230
    roll = None
    if local_attn_size != -1 and current_end > G and num_new + E > cache_size:      # :231-232
        evict = num_new + E - cache_size                                            # :235
        rolled = E - evict - sink_tokens                                            # :236
        local_end = E + current_end - G - evict                                     # :244-245
        local_start = local_end - num_new
        if rolled > 0:
            roll = (sink_tokens, sink_tokens + evict, rolled)                       # :257-260
        write_start = max(local_start, sink_tokens) if is_recompute else local_start  # :264
    else:
        local_end = E + current_end - G                                             # :293
        local_start = local_end - num_new
        write_start = max(local_start, sink_tokens) if is_recompute else local_start  # :302
        if sink_recache_after_switch:
            write_start = local_start                                               # :303-304
    roped_offset = max(0, write_start - local_start)
    write_len = max(0, local_end - write_start)
    if local_start < 0 or local_end > cache_size:
        raise RuntimeError(
            f"KV cache overflow: write window [{local_start}, {local_end}) outside cache of {cache_size} slots "
            f"(current_start={current_start}, n={num_new}, G={G}, E={E})")
    if sink_tokens > 0:                                                             # :331-353
        budget = max_attention_size - sink_tokens
        segments = [(0, sink_tokens)]
        if budget > 0:
            ws = max(sink_tokens, local_end - budget)
            if local_end > ws:
                segments.append((ws, local_end))
    else:                                                                           # :355-360
        segments = [(max(0, local_end - max_attention_size), local_end)]
    return KVPlan(current_end, is_recompute, roll, local_start, local_end, write_start, roped_offset, write_len,
                  segments, G if is_recompute else current_end, E if is_recompute else local_end)
