"""Interactive multi-prompt stream with KV-recache on prompt switch: host-side mirror of
pipeline/interactive_causal_inference.py::InteractiveCausalInferencePipeline (:20-432)."""
from __future__ import annotations

from typing import List

import torch

from .causal_inference import CausalInferencePipeline, _Profiler


class InteractiveCausalInferencePipeline(CausalInferencePipeline):
    def __init__(self, args, device, *, generator=None, text_encoder=None, vae=None):
        super().__init__(args, device, generator=generator, text_encoder=text_encoder, vae=vae)
        self.global_sink = getattr(args, "global_sink", False)

    def _reset_crossattn(self):
        for blk in self.crossattn_cache:
            blk["k"].zero_()
            blk["v"].zero_()
            blk["is_init"] = False

    def _recache_after_switch(self, output, current_start_frame, new_conditional_dict):
        """interactive_causal_inference.py:34-106: (optionally) zero the KV cache -- end indices are deliberately NOT
        reset (:43-44) --, re-encode the last <= local_attn_size generated frames in ONE forward at
        t = context_noise under the new prompt, and make the next forward rebuild the cross-attention K/V."""
        self._join_context()          # the last block's context pass (aux stream) must be done before the caches are zeroed / rewritten
        if not self.global_sink:
            for cache in self.kv_cache1:
                cache["k"].zero_()
                cache["v"].zero_()
        self._reset_crossattn()
        if current_start_frame == 0:
            return
        n = current_start_frame if self.local_attn_size == -1 else min(self.local_attn_size, current_start_frame)
        start = current_start_frame - n
        frames = output[:, start:current_start_frame]
        B = frames.shape[0]
        ctx_t = self._timestep(float(getattr(self.args, "context_noise", 0)), B, n, frames.device)
        # the reference also builds a flex-attention block mask here (:73-84); the KV-cache branch never reads it
        self.generator.model.block_mask = self.generator.model._prepare_blockwise_causal_attn_mask(
            device=frames.device, num_frames=n, frame_seqlen=self.frame_seq_length,
            num_frame_per_block=self.num_frame_per_block, local_attn_size=self.local_attn_size)
        self.generator(noisy_image_or_video=frames, conditional_dict=new_conditional_dict, timestep=ctx_t,
                       kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache,
                       current_start=start * self.frame_seq_length, sink_recache_after_switch=not self.global_sink, **self._kv_only_kw())
        self._reset_crossattn()

    @torch.no_grad()
    def inference(self, noise: torch.Tensor, *, text_prompts_list: List[List[str]], switch_frame_indices: List[int],
                  return_latents: bool = False, low_memory: bool = False, profile: bool = False):
        """Switch to prompt segment i+1 at the first block whose start frame >= switch_frame_indices[i] (:237)."""
        batch_size, num_output_frames = noise.shape[:2]
        assert len(text_prompts_list) >= 1, "text_prompts_list must not be empty"
        assert len(switch_frame_indices) == len(text_prompts_list) - 1, (
            "length of switch_frame_indices should be one less than text_prompts_list")
        assert num_output_frames % self.num_frame_per_block == 0
        num_blocks = num_output_frames // self.num_frame_per_block
        cond_list = [self._encode(p) for p in text_prompts_list]
        output = torch.zeros_like(noise)

        prof = _Profiler(profile)
        prof.start("init")
        self._setup(noise, num_output_frames)
        prof.stop("init")
        prof.start("diffusion")
        seg, start, switch_blocks = 0, 0, []
        next_switch = switch_frame_indices[0] if switch_frame_indices else None
        side = self._side_decoder(noise)
        for blk in range(num_blocks):
            nf = self.num_frame_per_block
            prof.block_start()
            if next_switch is not None and start >= next_switch:
                seg += 1
                self._recache_after_switch(output, start, cond_list[seg])
                switch_blocks.append(blk)
                next_switch = switch_frame_indices[seg] if seg < len(switch_frame_indices) else None
            cond = cond_list[seg]
            denoised = self._denoise_block(noise[:, start:start + nf], cond, start, batch_size, nf)
            output[:, start:start + nf] = denoised
            self._clean_context_pass(denoised, cond, start)
            if side is not None:
                side.push(denoised)
            prof.block_end()
            start += nf
        self._join_context()
        prof.stop("diffusion")
        prof.start("vae")
        video = None
        if side is not None:
            video = side.finish()
        elif self.vae is not None:
            video = self.vae.decode_to_pixel(output, use_cache=False)
            video = (video * 0.5 + 0.5).clamp(0, 1)
        prof.stop("vae")
        self.last_profile = prof.report(self.num_frame_per_block, switch_blocks=tuple(switch_blocks))
        if return_latents:
            return video, output
        return video
