"""Training-time reuse of the KV-cache forward (second half): host-side mirrors of the reference's
roll-out pipelines pipeline/streaming_training.py::StreamingTrainingPipeline and
pipeline/streaming_switch_training.py::StreamingSwitchTrainingPipeline -- chunk-by-chunk generation on a persistent KV cache
(`generate_chunk_with_cache`), a random exit step per block, context-noise re-encoding of every block, and the mid-chunk prompt
switch with KV recache (`_recache_after_switch`, :244-317).

FORWARD ONLY.  The reference runs the exit step of the blocks after `start_gradient_frame_index` under torch.enable_grad() for the
DMD loss; the HIP generator is an inference path (no backward), so `requires_grad=True` raises: what is reusable at training time
is every roll-out that the reference itself runs without gradients (`requires_grad=False`: the warm-up chunks, the frames before
the gradient window, evaluation roll-outs).  Same constructor arguments, method names, return values and cache geometry
((local_attn_size + slice_last_frames) frames) as the reference, so a trainer can hand these the HIP generator for its
no-grad roll-outs.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


class StreamingTrainingPipeline:
    def __init__(self, denoising_step_list: List[int], scheduler, generator, num_frame_per_block: int = 3,
                 same_step_across_blocks: bool = False, last_step_only: bool = False, context_noise: int = 0, **kwargs):
        self.scheduler = scheduler
        self.generator = generator
        self.denoising_step_list = denoising_step_list
        if self.denoising_step_list[-1] == 0:                      # streaming_training.py:33-34
            self.denoising_step_list = self.denoising_step_list[:-1]
        cfg = getattr(getattr(generator, "model", None), "cfg", None)
        self.num_transformer_blocks = cfg.num_layers if cfg is not None else 30        # reference hard-codes 30 / 1560 / 12 x 128
        self.frame_seq_length = cfg.frame_seqlen if cfg is not None else 1560
        self.num_heads = cfg.num_heads if cfg is not None else 12
        self.head_dim = cfg.head_dim if cfg is not None else 128
        self.text_len = cfg.text_len if cfg is not None else 512
        self.num_frame_per_block = num_frame_per_block
        self.context_noise = context_noise
        self.kv_cache1 = None
        self.crossattn_cache = None
        self.same_step_across_blocks = same_step_across_blocks
        self.last_step_only = last_step_only
        self.local_attn_size = kwargs.get("local_attn_size", -1)
        slice_last_frames = int(kwargs.get("slice_last_frames", 21))
        self.kv_cache_size = (self.local_attn_size + slice_last_frames) * self.frame_seq_length       # :49-50
        self.randn_like = torch.randn_like          # tests inject a counter-hash RNG (the reference calls torch.randn_like)
        self.randint = torch.randint

    # ---- streaming_training.py:54-71 ---------------------------------------------------------------------------------
    def generate_and_sync_list(self, num_blocks, num_denoising_steps, device):
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_initialized() else 0
        if rank == 0:
            indices = self.randint(low=0, high=num_denoising_steps, size=(num_blocks,), device=device)
            if self.last_step_only:
                indices = torch.ones_like(indices) * (num_denoising_steps - 1)
        else:
            indices = torch.empty(num_blocks, dtype=torch.long, device=device)
        if dist.is_initialized():
            dist.broadcast(indices, src=0)
        return indices.tolist()

    def _call(self, x, cond, timestep, start_frame, **kw):
        return self.generator(noisy_image_or_video=x, conditional_dict=cond, timestep=timestep, kv_cache=self.kv_cache1,
                              crossattn_cache=self.crossattn_cache, current_start=start_frame * self.frame_seq_length, **kw)

    def _step_tensor(self, value, batch, frames, device):
        return torch.ones([batch, frames], device=device, dtype=torch.int64) * value

    def _exit_info(self, exit_flags, return_sim_step, output):
        """streaming_training.py:228-246: the (from, to) timesteps of the exit step when it is shared by all blocks."""
        if not self.same_step_across_blocks:
            t_from, t_to = None, None
        else:
            ts = self.scheduler.timesteps
            steps = self.denoising_step_list

            def idx(v):
                return 1000 - torch.argmin((ts.to(torch.float32).cpu() - torch.as_tensor(v, dtype=torch.float32).cpu()).abs(), dim=0).item()
            if exit_flags[0] == len(steps) - 1:
                t_to, t_from = 0, idx(steps[exit_flags[0]])
            else:
                t_to, t_from = idx(steps[exit_flags[0] + 1]), idx(steps[exit_flags[0]])
        if return_sim_step:
            return output, t_from, t_to, exit_flags[0] + 1
        return output, t_from, t_to

    def _check_no_grad(self, requires_grad: bool):
        if requires_grad:
            raise NotImplementedError(
                "longlive_amd: the HIP generator is forward-only -- generate_chunk_with_cache(requires_grad=True) (the DMD loss's "
                "gradient window, streaming_training.py:195-209) needs the reference generator; pass requires_grad=False for the "
                "roll-outs the reference runs under torch.no_grad()")

    def _prepare(self):
        self.generator.model.local_attn_size = int(self.local_attn_size)
        self._set_all_modules_max_attention_size(int(self.local_attn_size))

    def _run_block(self, noisy_input, cond, exit_step: int, start_frame: int):
        """One block: no-grad denoising steps with re-noising up to and including the exit step (:140-209), then the
        context-noise re-encoding that refreshes the block's K/V (:211-233).  Returns denoised_pred."""
        B, nf = noisy_input.shape[:2]
        dev = noisy_input.device
        denoised = None
        for step_idx, current_timestep in enumerate(self.denoising_step_list):
            timestep = self._step_tensor(current_timestep, B, nf, dev)
            _, denoised = self._call(noisy_input, cond, timestep, start_frame)
            if step_idx == exit_step:
                break
            if step_idx < len(self.denoising_step_list) - 1:
                nxt = self.denoising_step_list[step_idx + 1]
                flat = denoised.flatten(0, 1)
                noisy_input = self.scheduler.add_noise(flat, self.randn_like(flat),
                                                       nxt * torch.ones([B * nf], device=dev, dtype=torch.long)
                                                       ).unflatten(0, denoised.shape[:2])
        context_timestep = torch.ones_like(timestep) * self.context_noise
        flat = denoised.flatten(0, 1)
        context_noisy = self.scheduler.add_noise(flat, self.randn_like(flat), context_timestep.flatten(0, 1)
                                                 ).unflatten(0, denoised.shape[:2])
        kw = {"kv_only": True} if getattr(self.generator, "supports_kv_only", False) else {}
        self._call(context_noisy, cond, context_timestep, start_frame, **kw)
        return denoised

    # ---- streaming_training.py:73-257 --------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_chunk_with_cache(self, noise: torch.Tensor, conditional_dict: dict, *, current_start_frame: int = 0,
                                  requires_grad: bool = True, return_sim_step: bool = False):
        self._check_no_grad(requires_grad)
        batch_size, chunk_frames = noise.shape[:2]
        assert chunk_frames % self.num_frame_per_block == 0
        num_blocks = chunk_frames // self.num_frame_per_block
        output = torch.zeros_like(noise)
        exit_flags = self.generate_and_sync_list(num_blocks, len(self.denoising_step_list), device=noise.device)
        self._prepare()
        local_start = 0
        for block_index in range(num_blocks):
            nf = self.num_frame_per_block
            exit_step = exit_flags[0] if self.same_step_across_blocks else exit_flags[block_index]
            denoised = self._run_block(noise[:, local_start:local_start + nf], conditional_dict, exit_step,
                                       current_start_frame + local_start)
            output[:, local_start:local_start + nf] = denoised
            local_start += nf
        return self._exit_info(exit_flags, return_sim_step, output)

    # ---- caches (streaming_training.py:259-311): the reference's geometry; end indices as host ints (no .item() syncs) ----
    def _initialize_kv_cache(self, batch_size, dtype, device):
        shape = [batch_size, self.kv_cache_size, self.num_heads, self.head_dim]
        self.kv_cache1 = [dict(k=torch.zeros(shape, dtype=dtype, device=device), v=torch.zeros(shape, dtype=dtype, device=device),
                               global_end_index=0, local_end_index=0) for _ in range(self.num_transformer_blocks)]

    def _initialize_crossattn_cache(self, batch_size, dtype, device):
        shape = [batch_size, self.text_len, self.num_heads, self.head_dim]
        self.crossattn_cache = [dict(k=torch.zeros(shape, dtype=dtype, device=device), v=torch.zeros(shape, dtype=dtype, device=device),
                                     is_init=False) for _ in range(self.num_transformer_blocks)]

    def clear_kv_cache(self):
        for blk in self.kv_cache1 or []:
            blk["k"].zero_()
            blk["v"].zero_()
            for key in ("global_end_index", "local_end_index"):
                if torch.is_tensor(blk.get(key)):
                    blk[key].zero_()
                else:
                    blk[key] = 0
        for blk in self.crossattn_cache or []:
            blk["k"].zero_()
            blk["v"].zero_()
            blk["is_init"] = False

    def _set_all_modules_max_attention_size(self, local_attn_size_value: int):
        if isinstance(local_attn_size_value, (list, tuple)):
            raise ValueError("_set_all_modules_max_attention_size expects an int, got list/tuple.")
        target = 32760 if int(local_attn_size_value) == -1 else int(local_attn_size_value) * self.frame_seq_length
        model = self.generator.model
        if hasattr(model, "max_attention_size"):
            model.max_attention_size = target
        for _, module in model.named_modules():
            if hasattr(module, "max_attention_size"):
                module.max_attention_size = target


class StreamingSwitchTrainingPipeline(StreamingTrainingPipeline):
    """Mid-chunk prompt switch: at the first block whose start >= switch_frame_index the KV cache is (optionally) zeroed and
    re-encoded from the last <= 21 generated frames under the new prompt, the cross-attention caches are reset, and generation
    continues with the second prompt (streaming_switch_training.py:18-243)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # the reference reads getattr(args, "global_sink", False) on the positional-args TUPLE, i.e. always False (:33)
        self.global_sink = False

    @torch.no_grad()
    def generate_chunk_with_cache(self, noise: torch.Tensor, conditional_dict: dict, *, current_start_frame: int = 0,
                                  requires_grad: bool = True, switch_frame_index: Optional[int] = None,
                                  switch_conditional_dict: Optional[dict] = None,
                                  switch_recache_frames: Optional[torch.Tensor] = None, return_sim_step: bool = False):
        if switch_conditional_dict is None or switch_frame_index is None:
            return super().generate_chunk_with_cache(noise=noise, conditional_dict=conditional_dict,
                                                     current_start_frame=current_start_frame, requires_grad=requires_grad,
                                                     return_sim_step=return_sim_step)
        self._check_no_grad(requires_grad)
        batch_size, chunk_frames = noise.shape[:2]
        assert chunk_frames % self.num_frame_per_block == 0
        num_blocks = chunk_frames // self.num_frame_per_block
        output = torch.zeros_like(noise)
        exit_flags = self.generate_and_sync_list(num_blocks, len(self.denoising_step_list), device=noise.device)
        self._prepare()
        local_start, using_second, cond = 0, False, conditional_dict
        for block_index in range(num_blocks):
            nf = self.num_frame_per_block
            if not using_second and local_start >= switch_frame_index:
                self._recache_after_switch(output[:, :local_start], current_start_frame + local_start, switch_conditional_dict,
                                           local_start, switch_recache_frames)
                cond, using_second = switch_conditional_dict, True
            exit_step = exit_flags[0] if self.same_step_across_blocks else exit_flags[block_index]
            denoised = self._run_block(noise[:, local_start:local_start + nf], cond, exit_step, current_start_frame + local_start)
            output[:, local_start:local_start + nf] = denoised
            local_start += nf
        return self._exit_info(exit_flags, return_sim_step, output)

    def _reset_crossattn(self):
        for blk in self.crossattn_cache:
            blk["k"].zero_()
            blk["v"].zero_()
            blk["is_init"] = False

    @torch.no_grad()
    def _recache_after_switch(self, output, current_start_frame, new_conditional_dict, local_start_frame=None,
                              switch_recache_frames=None):
        """streaming_switch_training.py:244-317.  Unlike the inference pipeline's recache it never sets
        sink_recache_after_switch, and it takes its frames from an external buffer of previous frames when one is given."""
        if not self.global_sink:
            for cache in self.kv_cache1:
                cache["k"].zero_()
                cache["v"].zero_()
        self._reset_crossattn()
        if current_start_frame == 0:
            return
        if switch_recache_frames is not None:
            frames = torch.cat([switch_recache_frames, output], dim=1)[:, -21:, ...]
        else:
            n = min(local_start_frame if local_start_frame is not None else current_start_frame, 21)
            frames = output[:, -n:]
        batch_size, num_recache_frames = frames.shape[:2]
        # the reference builds a flex-attention block mask here (:289-299); the KV-cache branch never reads it
        self.generator.model.block_mask = self.generator.model._prepare_blockwise_causal_attn_mask(
            device=frames.device, num_frames=num_recache_frames, frame_seqlen=self.frame_seq_length,
            num_frame_per_block=self.num_frame_per_block, local_attn_size=21)
        context_timestep = torch.ones([batch_size, num_recache_frames], device=frames.device, dtype=torch.int64) * self.context_noise
        kw = {"kv_only": True} if getattr(self.generator, "supports_kv_only", False) else {}
        self._call(frames, new_conditional_dict, context_timestep, current_start_frame - num_recache_frames, **kw)
        self._reset_crossattn()
