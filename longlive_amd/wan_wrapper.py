"""WanDiffusionWrapper for the MI355X path: same call signature and return values as the reference's
utils/wan_wrapper.py::WanDiffusionWrapper.forward (:224-300), so it can be handed to the reference's
CausalInferencePipeline / InteractiveCausalInferencePipeline through their `generator=` constructor argument
(pipeline/causal_inference.py:14-29) as well as to the pipelines in longlive_amd.pipeline.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .model import CausalWanModelHIP
from .scheduler import FlowMatchScheduler, uniform_value
from .synth import WanConfig, longlive_1_3b


class WanDiffusionWrapper(nn.Module):
    def __init__(self, model_name: str = "Wan2.1-T2V-1.3B", timestep_shift: float = 8.0, is_causal: bool = True,
                 local_attn_size: int = -1, sink_size: int = 0, *, cfg: Optional[WanConfig] = None, device="cuda",
                 state_dict: Optional[Dict[str, torch.Tensor]] = None):
        """Reference constructor arguments first (utils/wan_wrapper.py:121-128).  The reference loads
        `wan_models/{model_name}/` from disk; checkpoints are not available offline, so weights come from
        `state_dict` (reference names, with or without the `model.` prefix) or stay zero until load_state_dict."""
        super().__init__()
        if not is_causal:
            raise NotImplementedError("only the causal (KV-cache) generator is on the hot path")
        if cfg is None:
            if model_name != "Wan2.1-T2V-1.3B":
                raise NotImplementedError(f"unknown model {model_name}")
            cfg = longlive_1_3b(local_attn_size=local_attn_size, sink_size=sink_size)
        self.model = CausalWanModelHIP(cfg, device=device)
        self.model.eval()
        self.uniform_timestep = False
        self.scheduler = FlowMatchScheduler(shift=timestep_shift, sigma_min=0.0, extra_one_step=True)
        self.scheduler.set_timesteps(1000, training=True)
        self.seq_len = 1560 * local_attn_size if local_attn_size > 21 else 32760      # wan_wrapper.py:147
        if state_dict is not None:
            sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state_dict.items()}
            self.model.load_state_dict(sd, strict=True)

    def get_scheduler(self) -> FlowMatchScheduler:
        return self.scheduler

    supports_kv_only = True      # the pipelines pass kv_only=True for passes whose output they discard
    supports_layer_events = True  # ... and layer_wait / layer_record to run two forwards one layer apart on two streams

    @torch.no_grad()
    def forward(self, noisy_image_or_video: torch.Tensor, conditional_dict: dict, timestep: torch.Tensor,
                kv_cache: Optional[List[dict]] = None, crossattn_cache: Optional[List[dict]] = None,
                current_start: Optional[int] = None, classify_mode: Optional[bool] = False,
                concat_time_embeddings: Optional[bool] = False, clean_x: Optional[torch.Tensor] = None,
                aug_t: Optional[torch.Tensor] = None, cache_start: Optional[int] = None,
                sink_recache_after_switch: bool = False, kv_only: bool = False, layer_wait=None, layer_record=None):
        """noisy [B,F,16,H,W], timestep [B,F] -> (flow_pred, pred_x0), both [B,F,16,H,W].  kv_only (not in the reference's
        signature; default off): only update the KV caches, return (None, None) -- see CausalWanModelHIP.forward_frames."""
        if kv_cache is None or classify_mode or clean_x is not None:
            raise NotImplementedError("only the KV-cache inference call is implemented (SURVEY.md section 8a)")
        prompt_embeds = conditional_dict["prompt_embeds"]
        dev = self.model.patch_embedding.weight.device
        x = noisy_image_or_video.to(dev)
        t = timestep.to(dev)
        # a timestep tensor built by our pipelines from ONE host value says so (pipeline/causal_inference.py::_timestep): what
        # depends on the value alone -- sigma, time embedding, modulation table -- is then taken from memos
        t_uniform = uniform_value(timestep)
        sigma = None if kv_only else self.scheduler.sigma_of(t, uniform_value=t_uniform)       # wan_wrapper.py:195-197
        out = self.model.forward_frames(x, t, prompt_embeds.to(dev), kv_cache, crossattn_cache,
                                        int(current_start or 0), sink_recache_after_switch, sigma=sigma,
                                        kv_only=kv_only, layer_wait=layer_wait, layer_record=layer_record,
                                        t_uniform=t_uniform)
        if kv_only:
            return None, None
        flow, x0 = out
        dt = noisy_image_or_video.dtype
        return flow.to(dt), x0.to(dt)
