"""Throughput mode for one GPU: several independent prompt streams (BASELINE config 5: `num_samples` prompts per GPU,
configs/longlive_inference.yaml:23, inference.py:193-195) interleaved on separate HIP streams of ONE process.

Each stream is an ordinary CausalInferencePipeline with its own KV / cross-attention caches; all of them share one generator
(one copy of the weights).  A stream's kernels are issued on its own HIP stream, block by block, round robin from one Python
thread; the GPU then always has launches of both streams queued and runs them side by side whenever the resources allow: one
stream's row kernels, epilogues and the 28 CUs its 228-workgroup self-attention leaves idle are filled by the other stream's
dense kernels (two PROCESSES on one card measured +7 % in round 2: DESIGN.md section 7).  Nothing is shared between the
streams but read-only weights, so every stream's latents are bit-identical to the same stream run alone
(tests/test_model_gpu.py::test_interleaved_streams_are_bit_identical_to_solo_runs).
"""
from __future__ import annotations

from typing import Iterator, List, Sequence, Tuple

import torch


class InterleavedStreams:
    def __init__(self, pipelines: Sequence, device=None):
        assert len(pipelines) >= 1
        self.pipelines = list(pipelines)
        dev = torch.device(device) if device is not None else None
        self.streams: List[torch.cuda.Stream] = [torch.cuda.Stream(device=dev) for _ in self.pipelines]

    def stream(self, noises: Sequence[torch.Tensor], prompts: Sequence, outputs: Sequence = None) -> Iterator[List[Tuple[int, torch.Tensor]]]:
        """Yields, once per autoregressive block, [(start_frame, denoised_latents) per stream] as soon as the block's forwards
        of EVERY stream have been queued (nothing synchronises with the device).  The tensors of stream i belong to HIP stream
        self.streams[i]: consume them under `torch.cuda.stream(self.streams[i])`, or call join() first."""
        assert len(noises) == len(prompts) == len(self.pipelines)
        caller = torch.cuda.current_stream(noises[0].device)
        gens = []
        for i, (pipe, hs) in enumerate(zip(self.pipelines, self.streams)):
            hs.wait_stream(caller)                       # inputs prepared on the caller's stream
            with torch.cuda.stream(hs):
                gens.append(pipe.stream(noises[i], prompts[i], output=None if outputs is None else outputs[i]))
        alive, first = True, True
        while alive:
            step = []
            for i, (g, hs) in enumerate(zip(gens, self.streams)):
                if first and i > 0:
                    # stream 0's first block builds the state the streams share read-only (packed QKV weights, RoPE tables,
                    # LDS attributes): the others start behind it, once
                    hs.wait_stream(self.streams[0])
                with torch.cuda.stream(hs):
                    try:
                        step.append(next(g))
                    except StopIteration:
                        alive = False
            first = False
            if alive:
                yield step

    def join(self):
        """The caller's stream waits for everything queued on the member streams."""
        cur = torch.cuda.current_stream(self.streams[0].device)
        for hs in self.streams:
            cur.wait_stream(hs)

    @torch.no_grad()
    def inference(self, noises: Sequence[torch.Tensor], prompts: Sequence) -> List[torch.Tensor]:
        """Latents [B, T, 16, h, w] of every stream (the streams' inference(..., return_latents=True) results)."""
        outs = [torch.zeros_like(n) for n in noises]
        for _ in self.stream(noises, prompts, outputs=outs):
            pass
        self.join()
        return outs
