"""Frame-level autoregressive inference loop on MI355X: host-side mirror of
pipeline/causal_inference.py::CausalInferencePipeline (same constructor and `inference(...)` signature, :14-63).

Per block of `num_frame_per_block` latent frames: the warped denoising schedule (4 forwards, re-noising between
them), then one clean-context forward at t = context_noise that overwrites the block's K/V with clean-frame K/V
(:145-200).  All forwards go through the HIP generator; nothing here synchronises with the device except the
optional profiler.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from ..scheduler import tag_uniform
from ..wan_wrapper import WanDiffusionWrapper


class CausalInferencePipeline(nn.Module):
    def __init__(self, args, device, generator=None, text_encoder=None, vae=None):
        super().__init__()
        mk = getattr(args, "model_kwargs", None)
        kw = dict(vars(mk)) if (mk is not None and hasattr(mk, "__dict__")) else dict(mk or {})
        self.generator = WanDiffusionWrapper(**kw, is_causal=True, device=device) if generator is None else generator
        # text encoder (umT5) and VAE are outside the hot path: injected, never built here
        self.text_encoder = text_encoder
        self.vae = vae

        self.scheduler = self.generator.get_scheduler()
        steps = torch.tensor(list(args.denoising_step_list), dtype=torch.long)
        if args.warp_denoising_step:                                                   # causal_inference.py:35-37
            timesteps = torch.cat((self.scheduler.timesteps.cpu(), torch.tensor([0], dtype=torch.float32)))
            steps = timesteps[1000 - steps]
        self.denoising_step_list = steps
        # python floats: timestep tensors are built from host values, no device round trip
        self._step_values = [float(s) for s in steps]

        cfg = getattr(self.generator.model, "cfg", None)
        self.num_transformer_blocks = cfg.num_layers if cfg is not None else 30         # reference hard-codes 30 / 1560
        self.frame_seq_length = cfg.frame_seqlen if cfg is not None else 1560
        self.num_heads = cfg.num_heads if cfg is not None else 12
        self.head_dim = cfg.head_dim if cfg is not None else 128
        self.text_len = cfg.text_len if cfg is not None else 512

        self.kv_cache1 = None
        self.crossattn_cache = None
        self.args = args
        self.num_frame_per_block = getattr(args, "num_frame_per_block", 1)
        self.local_attn_size = _mk(args, "local_attn_size", -1)
        if self.num_frame_per_block > 1:
            self.generator.model.num_frame_per_block = self.num_frame_per_block
        self.randn_like = torch.randn_like      # re-noise source (causal_inference.py:175); tests inject a hash RNG
        self.last_profile = None
        self.verbose = False
        # Two HIP streams: a block's clean-context pass (output discarded, only refreshes the KV cache) runs on `_aux` while the
        # NEXT block's first denoising forward -- whose input is fresh noise -- runs on the main stream one layer behind it
        # (per-layer events: layer i of the follower waits until the context pass has left layer i's cache).  Same kernels on
        # the same data in the same per-cache order: bit-identical results (tests/test_model_gpu.py).  One stream's low-power
        # phases (epilogues, row kernels, launch ramps) and the CUs its 228-workgroup attention leaves idle then overlap the other's
        # dense kernels: +1.55 % frames/s with round 4's kernels (profiles/r04_ab_overlap_context.md).
        # ON by default since round 4 (generators that cannot run kv_only with per-layer events take the one-stream path);
        # `overlap_context = False` -- or LL_OVERLAP=0 for bench.py -- gives the one-stream schedule, which is what bench.py's
        # per-kernel table and roofline figure are measured on (kernels running alone on the device).
        self.overlap_context = True
        self.overlap_decode = False      # inference(): decode each block on a second stream while the next one is generated (same video, bit for bit)
        self._aux = None
        self._ctx_events = None
        self._timestep_memo = {}
        self._ctx_pending = False

    # ------------------------------------------------------------------------------------------------------------
    def _encode(self, text_prompts):
        if isinstance(text_prompts, dict):
            return text_prompts
        if torch.is_tensor(text_prompts):
            return {"prompt_embeds": text_prompts}
        if self.text_encoder is None:
            raise RuntimeError("no text_encoder was injected: pass prompt embeddings ({'prompt_embeds': [B,512,4096]}) "
                               "or construct the pipeline with text_encoder=")
        return self.text_encoder(text_prompts=text_prompts)

    def _timestep(self, value: float, batch: int, frames: int, device):
        """[batch, frames] filled with ONE host value (causal_inference.py:166-170,193 build theirs the same way).  The tensor says
        so (`_ll_uniform_value`): our wrapper / scheduler then take what depends on the value alone -- sigma, the time embedding and
        the modulation table of all layers -- from memos instead of seven launches per forward.  One tensor per (value, shape,
        device, HIP stream), never written to."""
        dev = torch.device(device)
        key = (float(value), batch, frames, str(dev), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        t = self._timestep_memo.get(key)
        if t is None:
            t = tag_uniform(torch.full([batch, frames], value, dtype=torch.float32, device=dev), value)
            self._timestep_memo[key] = t
        return t

    def _denoise_block(self, noisy_input, cond, start_frame: int, batch_size: int, nframes: int):
        """4-step denoise + re-noise of one block (causal_inference.py:154-188).  Returns denoised_pred."""
        dev = noisy_input.device
        cs = start_frame * self.frame_seq_length
        denoised = None
        for index, tval in enumerate(self._step_values):
            timestep = self._timestep(tval, batch_size, nframes, dev)
            kw = {}
            if self._ctx_pending:        # the previous block's context pass is still running on the aux stream
                kw["layer_wait"] = self._ctx_events
                self._ctx_pending = False      # later forwards are ordered behind this one on the main stream
            _, denoised = self.generator(noisy_image_or_video=noisy_input, conditional_dict=cond, timestep=timestep,
                                         kv_cache=self.kv_cache1, crossattn_cache=self.crossattn_cache, current_start=cs, **kw)
            if index < len(self._step_values) - 1:
                nxt = self._timestep(self._step_values[index + 1], batch_size * nframes, 1, dev)      # [B*F, 1]: add_noise flattens it
                flat = denoised.flatten(0, 1)
                noisy_input = self.scheduler.add_noise(flat, self.randn_like(flat), nxt).unflatten(0, denoised.shape[:2])
        return denoised

    def _kv_only_kw(self) -> dict:
        """The output of the clean-context / recache pass is discarded (causal_inference.py:192-200): a generator that can
        stop after the last layer's K/V insert is told so.  Any other generator gets the reference's exact call."""
        return {"kv_only": True} if getattr(self.generator, "supports_kv_only", False) else {}

    def _clean_context_pass(self, denoised, cond, start_frame: int):
        """Re-run at t = context_noise so the cache holds clean-frame K/V (causal_inference.py:192-200)."""
        B, nf = denoised.shape[:2]
        ctx_t = self._timestep(float(getattr(self.args, "context_noise", 0)), B, nf, denoised.device)
        if self._use_overlap(denoised):
            main = torch.cuda.current_stream(denoised.device)
            if self._aux is None:
                self._aux = torch.cuda.Stream(device=denoised.device)
                self._ctx_events = [torch.cuda.Event() for _ in range(self.num_transformer_blocks)]
            self._aux.wait_stream(main)             # the denoised latents (and this block's K/V inserts) are ready
            denoised.record_stream(self._aux)       # allocated on the main stream, read by the aux stream
            ctx_t.record_stream(self._aux)
            with torch.cuda.stream(self._aux):
                self.generator(noisy_image_or_video=denoised, conditional_dict=cond, timestep=ctx_t, kv_cache=self.kv_cache1,
                               crossattn_cache=self.crossattn_cache, current_start=start_frame * self.frame_seq_length,
                               layer_record=self._ctx_events, **self._kv_only_kw())
            self._ctx_pending = True
            return
        self.generator(noisy_image_or_video=denoised, conditional_dict=cond, timestep=ctx_t, kv_cache=self.kv_cache1,
                       crossattn_cache=self.crossattn_cache, current_start=start_frame * self.frame_seq_length,
                       **self._kv_only_kw())

    def _side_decoder(self, noise: torch.Tensor):
        if self.overlap_decode and self.vae is not None and noise.is_cuda and noise.shape[0] == 1:
            return _SideDecoder(self.vae)
        return None

    def _use_overlap(self, t: torch.Tensor) -> bool:
        return (self.overlap_context and t.is_cuda and getattr(self.generator, "supports_layer_events", False)
                and getattr(self.generator, "supports_kv_only", False))

    def _join_context(self):
        """Make the main stream wait for an outstanding context pass (before anything but the next block's first forward
        touches the caches: a recache, the end of a run, a caller that reads the caches)."""
        if self._aux is not None and self._ctx_pending:
            torch.cuda.current_stream(self._aux.device).wait_stream(self._aux)
        self._ctx_pending = False

    def _setup(self, noise, num_output_frames):
        self._join_context()          # a previous run's last context pass must not outlive its caches
        local_attn_cfg = _mk(self.args, "local_attn_size", -1)
        kv_cache_size = (local_attn_cfg if local_attn_cfg != -1 else num_output_frames) * self.frame_seq_length
        if self.verbose:
            print(f"kv_cache_size: {kv_cache_size} (frame_seq_length: {self.frame_seq_length}, "
                  f"num_output_frames: {num_output_frames})")
        self._initialize_kv_cache(noise.shape[0], noise.dtype, noise.device, kv_cache_size_override=kv_cache_size)
        self._initialize_crossattn_cache(noise.shape[0], noise.dtype, noise.device)
        self.generator.model.local_attn_size = self.local_attn_size
        self._set_all_modules_max_attention_size(self.local_attn_size)

    # ------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def stream(self, noise: torch.Tensor, text_prompts, output: Optional[torch.Tensor] = None):
        """Generator form of `inference`: yields (start_frame, denoised_latents[B, F, 16, h, w]) after every
        autoregressive block, so frames can be consumed (decoded / displayed / timed) while generation goes on."""
        batch_size, num_output_frames = noise.shape[:2]
        assert num_output_frames % self.num_frame_per_block == 0
        cond = self._encode(text_prompts)
        self._setup(noise, num_output_frames)
        nf = self.num_frame_per_block
        for start in range(0, num_output_frames, nf):
            denoised = self._denoise_block(noise[:, start:start + nf], cond, start, batch_size, nf)
            if output is not None:
                output[:, start:start + nf] = denoised
            self._clean_context_pass(denoised, cond, start)
            yield start, denoised
        self._join_context()

    @torch.no_grad()
    def stream_video(self, noise: torch.Tensor, text_prompts, output: Optional[torch.Tensor] = None,
                     overlap_decode: bool = False):
        """Live form of `inference`: yields (start_frame, pixels[B, T', 3, H, W] in [0, 1]) after every block, decoding
        each block with the VAE's streaming cache (WanVAE_.cached_decode, wan/modules/vae.py:571-593) -- the first block
        gives 1 + 4 (F - 1) frames, later ones 4 F.  The concatenation equals `inference(...)`'s video bit for bit.

        overlap_decode: block i is decoded on a second HIP stream while block i + 1 is generated on the caller's stream (the
        decoder's staging-bound convolutions run beside the power-bound DiT kernels); block i's pixels are then yielded after
        block i + 1's generation has been queued, i.e. one block later, with the caller's stream made to wait for that decode.
        Same values, same order."""
        if self.vae is None:
            raise RuntimeError("stream_video needs a VAE (pass vae= to the pipeline)")
        self.vae.model.clear_cache()
        pending = None                                   # (start, pixels, event) of the block being decoded on the side stream
        side = None
        try:
            for start, latents in self.stream(noise, text_prompts, output=output):
                if not (overlap_decode and latents.is_cuda):
                    video = self.vae.decode_to_pixel(latents, use_cache=True)
                    yield start, (video * 0.5 + 0.5).clamp(0, 1)
                    continue
                main = torch.cuda.current_stream(latents.device)
                if side is None:
                    side = torch.cuda.Stream(device=latents.device)
                side.wait_stream(main)                   # the block's denoised latents are final (the context pass only reads them)
                latents.record_stream(side)
                with torch.cuda.stream(side):            # decodes are ordered among themselves by the side stream (streaming cache)
                    video = self.vae.decode_to_pixel(latents, use_cache=True)
                    video = (video * 0.5 + 0.5).clamp(0, 1)
                    ev = torch.cuda.Event()
                    ev.record(side)
                if pending is not None:
                    ps, pv, pe = pending
                    main.wait_event(pe)
                    pv.record_stream(main)
                    yield ps, pv
                pending = (start, video, ev)
            if pending is not None:
                ps, pv, pe = pending
                torch.cuda.current_stream(pv.device).wait_event(pe)
                pv.record_stream(torch.cuda.current_stream(pv.device))
                pending = None
                yield ps, pv
        finally:
            if side is not None:
                torch.cuda.current_stream(side.device).wait_stream(side)
            self.vae.model.clear_cache()

    @torch.no_grad()
    def inference(self, noise: torch.Tensor, text_prompts: List[str], return_latents: bool = False,
                  profile: bool = False, low_memory: bool = False):
        """noise [B, T, 16, H/8, W/8] -> video [B, T', 3, H, W] in [0,1] (None without a VAE) and, with
        return_latents, the denoised latents [B, T, 16, H/8, W/8]."""
        batch_size, num_output_frames = noise.shape[:2]
        assert num_output_frames % self.num_frame_per_block == 0
        num_blocks = num_output_frames // self.num_frame_per_block
        cond = self._encode(text_prompts)
        output = torch.zeros_like(noise)      # low_memory (CPU staging) is pointless with 288 GB of HBM: ignored

        prof = _Profiler(profile)
        prof.start("init")
        self._setup(noise, num_output_frames)
        prof.stop("init")
        prof.start("diffusion")
        start = 0
        side = self._side_decoder(noise)
        for _ in range(num_blocks):
            nf = self.num_frame_per_block
            prof.block_start()
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
        self.last_profile = prof.report(self.num_frame_per_block, switch_blocks=())
        if return_latents:
            return video, output
        return video

    # ---- cache allocation (causal_inference.py:255-293) ----------------------------------------------------------
    def _initialize_kv_cache(self, batch_size, dtype, device, kv_cache_size_override: Optional[int] = None):
        if kv_cache_size_override is not None:
            size = kv_cache_size_override
        else:
            size = self.local_attn_size * self.frame_seq_length if self.local_attn_size != -1 else 32760
        shape = [batch_size, size, self.num_heads, self.head_dim]
        self.kv_cache1 = [dict(k=torch.zeros(shape, dtype=dtype, device=device),
                               v=torch.zeros(shape, dtype=dtype, device=device),
                               global_end_index=0, local_end_index=0)       # host ints: no .item() syncs
                          for _ in range(self.num_transformer_blocks)]

    def _initialize_crossattn_cache(self, batch_size, dtype, device):
        shape = [batch_size, self.text_len, self.num_heads, self.head_dim]
        self.crossattn_cache = [dict(k=torch.zeros(shape, dtype=dtype, device=device),
                                     v=torch.zeros(shape, dtype=dtype, device=device), is_init=False)
                                for _ in range(self.num_transformer_blocks)]

    def _set_all_modules_max_attention_size(self, local_attn_size_value: int):
        """causal_inference.py:295-329."""
        target = 32760 if local_attn_size_value == -1 else int(local_attn_size_value) * self.frame_seq_length
        model = self.generator.model
        if hasattr(model, "max_attention_size"):
            model.max_attention_size = target
        for _, module in model.named_modules():
            if hasattr(module, "max_attention_size"):
                module.max_attention_size = target


def _mk(args, name, default):
    mk = getattr(args, "model_kwargs", None)
    if mk is None:
        return default
    if isinstance(mk, dict):
        return mk.get(name, default)
    return getattr(mk, name, default)


class _SideDecoder:
    """inference() with overlap_decode: every finished block goes through the VAE's streaming decode on a second HIP stream while
    the next block is generated; the pieces concatenated equal the one-shot decode of the finished latents bit for bit
    (tests/test_vae_gpu.py).  Batch 1 only (the streaming cache holds one stream)."""

    def __init__(self, vae):
        self.vae, self.pieces, self.side = vae, [], None
        vae.model.clear_cache()

    def push(self, latents: torch.Tensor):
        main = torch.cuda.current_stream(latents.device)
        if self.side is None:
            self.side = torch.cuda.Stream(device=latents.device)
        self.side.wait_stream(main)
        latents.record_stream(self.side)
        with torch.cuda.stream(self.side):
            v = self.vae.decode_to_pixel(latents, use_cache=True)
            self.pieces.append((v * 0.5 + 0.5).clamp(0, 1))

    def finish(self) -> Optional[torch.Tensor]:
        if self.side is None:                       # no block was pushed
            self.vae.model.clear_cache()
            return None
        main = torch.cuda.current_stream(self.side.device)
        main.wait_stream(self.side)
        for p in self.pieces:
            p.record_stream(main)
        video = torch.cat(self.pieces, 1)
        self.pieces = []
        self.vae.model.clear_cache()
        return video


class _Profiler:
    """HIP-event timing with the reference's report semantics (causal_inference.py:97-107,202-248): per-block
    times; steady state = blocks after the first, and (interactive) without the switch blocks."""

    def __init__(self, enabled: bool):
        self.enabled = enabled and torch.cuda.is_available()
        self.ev = {}
        self.blocks = []
        self._b0 = None

    def start(self, name):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.ev[name] = [e, None]

    def stop(self, name):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.ev[name][1] = e

    def block_start(self):
        if self.enabled:
            self._b0 = torch.cuda.Event(enable_timing=True)
            self._b0.record()

    def block_end(self):
        if self.enabled:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.blocks.append((self._b0, e))

    def report(self, frames_per_block: int, switch_blocks=()):
        if not self.enabled:
            return None
        torch.cuda.synchronize()
        times = {k: a.elapsed_time(b) for k, (a, b) in self.ev.items() if b is not None}
        bt = [a.elapsed_time(b) for a, b in self.blocks]
        steady = [t for i, t in enumerate(bt) if i > 0 and i not in set(switch_blocks)] or bt
        avg = sum(steady) / max(1, len(steady))
        rep = dict(times_ms=times, block_times_ms=bt, steady_block_ms=avg,
                   ms_per_latent_frame=avg / frames_per_block, switch_blocks=list(switch_blocks))
        if switch_blocks:
            sw = [bt[i] for i in switch_blocks if i < len(bt)]
            rep["switch_block_ms"] = sw
            rep["switch_latency_ms"] = [t - avg for t in sw]
        return rep
