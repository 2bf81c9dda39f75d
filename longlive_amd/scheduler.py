"""FlowMatchScheduler as used on the inference path (utils/scheduler.py:106-176 with the arguments of
utils/wan_wrapper.py:141-144: shift=timestep_shift, sigma_min=0, extra_one_step=True, set_timesteps(1000)).

The table is host data (1000 fp32 entries); add_noise runs in a HIP kernel on the device tables."""
from __future__ import annotations

import torch

from . import ops


def tag_uniform(t: torch.Tensor, value: float) -> torch.Tensor:
    """Mark `t` as filled with the one host value `value` (with the tensor's version counter: an in-place write voids the tag)."""
    t._ll_uniform_value = (float(value), t._version)
    return t


def uniform_value(t) -> "float | None":
    """The host value a tensor was tagged with by tag_uniform, or None (untagged, or written to since)."""
    tag = getattr(t, "_ll_uniform_value", None)
    if tag is None or tag[1] != t._version:
        return None
    return tag[0]


class FlowMatchScheduler:
    def __init__(self, shift: float = 5.0, sigma_min: float = 0.0, sigma_max: float = 1.0,
                 num_train_timesteps: int = 1000, num_inference_steps: int = 1000, extra_one_step: bool = True):
        self.shift, self.sigma_min, self.sigma_max = shift, sigma_min, sigma_max
        self.num_train_timesteps = num_train_timesteps
        self.extra_one_step = extra_one_step
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps: int = 1000, denoising_strength: float = 1.0, training: bool = False):
        """utils/scheduler.py:118-133 (fp32, CPU)."""
        start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        if self.extra_one_step:
            sig = torch.linspace(start, self.sigma_min, num_inference_steps + 1)[:-1]
        else:
            sig = torch.linspace(start, self.sigma_min, num_inference_steps)
        self.sigmas = self.shift * sig / (1 + (self.shift - 1) * sig)
        self.timesteps = self.sigmas * self.num_train_timesteps
        self._dev = {}
        self._sigma_memo = {}

    def tables(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.timesteps.to(device=device, dtype=torch.float32).contiguous(),
                              self.sigmas.to(device=device, dtype=torch.float32).contiguous())
        return self._dev[key]

    def sigma_of(self, timestep: torch.Tensor, uniform_value=None) -> torch.Tensor:
        """sigmas[argmin |timesteps - t|] on the device, float32 [numel].  uniform_value: the caller knows every entry equals this
        host value (the pipelines' timestep tensors): the lookup is done once per (value, numel, device, HIP stream)."""
        ts, sg = self.tables(timestep.device)
        if uniform_value is None or not timestep.is_cuda:
            return ops.sigma_lookup(timestep.reshape(-1).to(torch.float32).contiguous(), ts, sg)
        key = (float(uniform_value), timestep.numel(), str(timestep.device), torch.cuda.current_stream(timestep.device).cuda_stream)
        hit = self._sigma_memo.get(key)
        if hit is None:
            if len(self._sigma_memo) >= 256:
                self._sigma_memo = {}
            hit = self._sigma_memo[key] = ops.sigma_lookup(timestep.reshape(-1).to(torch.float32).contiguous(), ts, sg)
        return hit

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
        """utils/scheduler.py:159-176: (1 - sigma) * x0 + sigma * noise, per leading index, result in noise.dtype."""
        uniform = uniform_value(timestep)                            # (a view of the tensor would not carry the tag)
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        sigma = self.sigma_of(timestep.to(noise.device), uniform_value=uniform)
        return ops.add_noise(original_samples.to(torch.bfloat16).contiguous(), noise.to(torch.bfloat16).contiguous(),
                             sigma).type_as(noise)
