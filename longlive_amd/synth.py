"""Deterministic synthetic weights / inputs for LongLive-1.3B-shaped models.

There is no network (no checkpoints), so every config runs random-init weights.
`torch.randn` is not bit-reproducible across CPU ISAs / devices, so everything
here is derived from an integer counter hash (splitmix64 finaliser) evaluated
with int64 tensor ops: the same (seed, name, index) gives the same bf16 value
on the build container's CPU, on the GPU box's CPU and on the MI355X itself.

Init statistics follow the reference (wan/modules/causal_model.py:411,495,
1265-1287): xavier-uniform linears, N(0, .02) text/time embeddings,
modulation ~ N(0,1)/sqrt(dim).  Deliberate deviations, so that parity tests
exercise every term: biases are N(0, .02) instead of zero, norm weights are
1 + N(0, .1) instead of one, and head.head.weight is N(0, .02) instead of the
reference's zero init (:1287), which would make the output bias-only.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, Tuple

import torch

_M64 = (1 << 64) - 1
FORCE_TORCH_HASH = False       # tests: evaluate the hash with int64 tensor ops on the GPU too (the path ll_synth_hash replaces)
CPU_HASH = None                # optional accelerator for the CPU path, same integers (oracle/fast_hash.py installs a host build of
                               # csrc/synth_hash.h: the golden generators and the slow oracle tests hash billions of weights)


def _s64(x: int) -> int:
    """Python int -> two's complement int64 value."""
    x &= _M64
    return x - (1 << 64) if x >= (1 << 63) else x


_C1 = _s64(0xBF58476D1CE4E5B9)
_C2 = _s64(0x94D049BB133111EB)
_GOLD = _s64(0x9E3779B97F4A7C15)


def _lsr(x: torch.Tensor, n: int) -> torch.Tensor:
    return (x >> n) & ((1 << (64 - n)) - 1)


def _mix(x: torch.Tensor) -> torch.Tensor:
    x = (x ^ _lsr(x, 30)) * _C1
    x = (x ^ _lsr(x, 27)) * _C2
    return x ^ _lsr(x, 31)


def _stream(seed: int, name: str) -> int:
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return _s64((seed * 0x9E3779B97F4A7C15 + h * 0xD1B54A32D192ED03 + 0x632BE59BD9B4E019) & _M64)


def _counter(numel: int, device) -> torch.Tensor:
    return torch.arange(numel, dtype=torch.int64, device=device)


def _device_hash(kind: int, s: int, numel: int, device) -> torch.Tensor:
    """The same integers from the library's hash kernel (ll_synth_hash, csrc/synth_hash.h): on a GPU no int64 tensor op is issued
    (a python process under `rocprofv3 --pmc` dies in torch's int64 elementwise kernels on this image; with this, bench.py is
    profilable) and generation is one launch instead of ~40 per chunk."""
    from . import _lib
    out = torch.empty(numel, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.load().ll_synth_hash(out.data_ptr(), 0, numel, s & _M64, kind, torch.cuda.current_stream().cuda_stream), "ll_synth_hash")
    return out


def hash_uniform(seed: int, name: str, shape, device="cpu", chunk: int = 1 << 26) -> torch.Tensor:
    """float32 uniform in [0, 1) with 24 exact bits."""
    numel = int(math.prod(shape))
    s = _stream(seed, name)
    if torch.device(device).type == "cuda" and not FORCE_TORCH_HASH:
        return _device_hash(0, s, numel, device).view(*shape)
    if torch.device(device).type == "cpu" and CPU_HASH is not None and not FORCE_TORCH_HASH:
        return CPU_HASH(0, s, numel).view(*shape)
    out = torch.empty(numel, dtype=torch.float32, device=device)
    for lo in range(0, numel, chunk):
        hi = min(numel, lo + chunk)
        idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
        h = _mix(idx * _GOLD + s)
        out[lo:hi] = _lsr(h, 40).to(torch.float32) * (1.0 / (1 << 24))
    return out.view(*shape)


def hash_normal(seed: int, name: str, shape, device="cpu", chunk: int = 1 << 25) -> torch.Tensor:
    """float32 approx. N(0,1): Irwin-Hall sum of 12 uniforms of 16 bits (exact integer sum)."""
    numel = int(math.prod(shape))
    s = _stream(seed, name)
    if torch.device(device).type == "cuda" and not FORCE_TORCH_HASH:
        return _device_hash(1, s, numel, device).view(*shape)
    if torch.device(device).type == "cpu" and CPU_HASH is not None and not FORCE_TORCH_HASH:
        return CPU_HASH(1, s, numel).view(*shape)
    out = torch.empty(numel, dtype=torch.float32, device=device)
    for lo in range(0, numel, chunk):
        hi = min(numel, lo + chunk)
        idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
        acc = torch.zeros(hi - lo, dtype=torch.int64, device=device)
        for r in range(3):
            h = _mix((idx * 3 + r) * _GOLD + s)
            for k in range(4):
                acc += (h >> (16 * k)) & 0xFFFF
        # sum of 12 U{0..65535}: mean 12*32767.5, var 12*(65536^2-1)/12
        out[lo:hi] = (acc - 393210).to(torch.float32) * (1.0 / 65536.0)
    return out.view(*shape)


@dataclass
class WanConfig:
    """Architecture constants (wan/configs/wan_t2v_1_3B.py:17-27; causal_model.py:523-539)."""
    dim: int = 1536
    ffn_dim: int = 8960
    num_heads: int = 12
    num_layers: int = 30
    in_dim: int = 16
    out_dim: int = 16
    freq_dim: int = 256
    text_dim: int = 4096
    text_len: int = 512
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    eps: float = 1e-6
    local_attn_size: int = 12
    sink_size: int = 3
    # latent geometry (inference.py:193-195): 16 x 60 x 104 per frame
    lat_h: int = 60
    lat_w: int = 104

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads

    @property
    def frame_seqlen(self) -> int:
        return (self.lat_h // self.patch_size[1]) * (self.lat_w // self.patch_size[2])


def longlive_1_3b(**kw) -> WanConfig:
    return WanConfig(**kw)


def toy_config(**kw) -> WanConfig:
    """Small shape that keeps head_dim = 128 (the kernels' specialisation)."""
    base = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, text_len=16,
                lat_h=8, lat_w=12, local_attn_size=3, sink_size=1)
    base.update(kw)
    return WanConfig(**base)


def param_shapes(cfg: WanConfig) -> Dict[str, Tuple[int, ...]]:
    """State-dict names/shapes of CausalWanModel (module tree causal_model.py:90-95,395-411,491-495,599-619)."""
    d, f = cfg.dim, cfg.ffn_dim
    shapes: Dict[str, Tuple[int, ...]] = {
        "patch_embedding.weight": (d, cfg.in_dim, *cfg.patch_size),
        "patch_embedding.bias": (d,),
        "text_embedding.0.weight": (d, cfg.text_dim), "text_embedding.0.bias": (d,),
        "text_embedding.2.weight": (d, d), "text_embedding.2.bias": (d,),
        "time_embedding.0.weight": (d, cfg.freq_dim), "time_embedding.0.bias": (d,),
        "time_embedding.2.weight": (d, d), "time_embedding.2.bias": (d,),
        "time_projection.1.weight": (6 * d, d), "time_projection.1.bias": (6 * d,),
    }
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        shapes[p + "modulation"] = (1, 6, d)
        for att in ("self_attn", "cross_attn"):
            for lin in ("q", "k", "v", "o"):
                shapes[p + f"{att}.{lin}.weight"] = (d, d)
                shapes[p + f"{att}.{lin}.bias"] = (d,)
            shapes[p + f"{att}.norm_q.weight"] = (d,)
            shapes[p + f"{att}.norm_k.weight"] = (d,)
        shapes[p + "norm3.weight"] = (d,)
        shapes[p + "norm3.bias"] = (d,)
        shapes[p + "ffn.0.weight"] = (f, d); shapes[p + "ffn.0.bias"] = (f,)
        shapes[p + "ffn.2.weight"] = (d, f); shapes[p + "ffn.2.bias"] = (d,)
    out = math.prod(cfg.patch_size) * cfg.out_dim
    shapes["head.head.weight"] = (out, d)
    shapes["head.head.bias"] = (out,)
    shapes["head.modulation"] = (1, 2, d)
    return shapes


def synth_param(cfg: WanConfig, name: str, shape, seed: int = 0, device="cpu",
                dtype=torch.bfloat16) -> torch.Tensor:
    if name.endswith("modulation"):
        w = hash_normal(seed, name, shape, device) * (1.0 / math.sqrt(cfg.dim))
    elif name.endswith(".bias"):
        w = hash_normal(seed, name, shape, device) * 0.02
    elif "norm" in name and name.endswith(".weight"):
        w = 1.0 + hash_normal(seed, name, shape, device) * 0.1
    elif name.startswith(("text_embedding", "time_embedding", "head.head")):
        w = hash_normal(seed, name, shape, device) * 0.02
    else:  # xavier-uniform on weight.flatten(1)
        fan_out = shape[0]
        fan_in = int(math.prod(shape[1:]))
        a = math.sqrt(6.0 / (fan_in + fan_out))
        w = (hash_uniform(seed, name, shape, device) * 2.0 - 1.0) * a
    return w.to(dtype)


def synth_state_dict(cfg: WanConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16,
                     layers=None) -> Dict[str, torch.Tensor]:
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if layers is not None and name.startswith("blocks."):
            if int(name.split(".")[1]) not in layers:
                continue
        sd[name] = synth_param(cfg, name, shape, seed, device, dtype)
    return sd


def synth_noise(cfg: WanConfig, num_frames: int, seed: int = 0, batch: int = 1, device="cpu",
                dtype=torch.bfloat16, name: str = "noise") -> torch.Tensor:
    """[B, T, 16, H, W] ~ N(0,1) (inference.py:193-195)."""
    shape = (batch, num_frames, cfg.in_dim, cfg.lat_h, cfg.lat_w)
    return hash_normal(seed, name, shape, device).to(dtype)


def synth_prompt_embeds(cfg: WanConfig, seed: int = 1, batch: int = 1, valid_tokens: int | None = None,
                        device="cpu", dtype=torch.bfloat16) -> torch.Tensor:
    """[B, text_len, text_dim] ~ N(0,1) with the padded tail zeroed (utils/wan_wrapper.py:52-53)."""
    shape = (batch, cfg.text_len, cfg.text_dim)
    x = hash_normal(seed, "prompt_embeds", shape, device)
    if valid_tokens is None:
        valid_tokens = min(40, cfg.text_len)
    x[:, valid_tokens:] = 0
    return x.to(dtype)


# ---------------------------------------------------------------------------------------------------------------------
# Wan VAE decoder (wan/modules/vae.py:369-472, 612-624: dim 96, z_dim 16, dim_mult [1,2,4,4], 2 res blocks,
# temporal upsample [True, True, False])
@dataclass
class VaeConfig:
    dim: int = 96
    z_dim: int = 16
    dim_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    temporal_upsample: Tuple[bool, ...] = (True, True, False)


def vae_decoder_layout(cfg: VaeConfig):
    """The decoder's layer list in execution order: ('res', name, cin, cout) | ('attn', name, c) |
    ('up3d' | 'up2d', name, c).  Mirrors Decoder3d.__init__ (vae.py:386-421)."""
    dims = [cfg.dim * u for u in [cfg.dim_mult[-1]] + list(cfg.dim_mult[::-1])]
    layers = [("res", "decoder.middle.0", dims[0], dims[0]), ("attn", "decoder.middle.1", dims[0]),
              ("res", "decoder.middle.2", dims[0], dims[0])]
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(cfg.num_res_blocks + 1):
            layers.append(("res", f"decoder.upsamples.{idx}", cin, cout))
            idx += 1
            cin = cout
        if i != len(cfg.dim_mult) - 1:
            layers.append(("up3d" if cfg.temporal_upsample[i] else "up2d", f"decoder.upsamples.{idx}", cout))
            idx += 1
    return dims, layers


def vae_decoder_param_shapes(cfg: VaeConfig) -> Dict[str, Tuple[int, ...]]:
    dims, layers = vae_decoder_layout(cfg)
    sh: Dict[str, Tuple[int, ...]] = {"conv2.weight": (cfg.z_dim, cfg.z_dim, 1, 1, 1), "conv2.bias": (cfg.z_dim,),
                                      "decoder.conv1.weight": (dims[0], cfg.z_dim, 3, 3, 3), "decoder.conv1.bias": (dims[0],)}
    for L in layers:
        kind, name = L[0], L[1]
        if kind == "res":
            cin, cout = L[2], L[3]
            sh[name + ".residual.0.gamma"] = (cin, 1, 1, 1)
            sh[name + ".residual.2.weight"] = (cout, cin, 3, 3, 3); sh[name + ".residual.2.bias"] = (cout,)
            sh[name + ".residual.3.gamma"] = (cout, 1, 1, 1)
            sh[name + ".residual.6.weight"] = (cout, cout, 3, 3, 3); sh[name + ".residual.6.bias"] = (cout,)
            if cin != cout:
                sh[name + ".shortcut.weight"] = (cout, cin, 1, 1, 1); sh[name + ".shortcut.bias"] = (cout,)
        elif kind == "attn":
            c = L[2]
            sh[name + ".norm.gamma"] = (c, 1, 1)
            sh[name + ".to_qkv.weight"] = (3 * c, c, 1, 1); sh[name + ".to_qkv.bias"] = (3 * c,)
            sh[name + ".proj.weight"] = (c, c, 1, 1); sh[name + ".proj.bias"] = (c,)
        else:
            c = L[2]
            sh[name + ".resample.1.weight"] = (c // 2, c, 3, 3); sh[name + ".resample.1.bias"] = (c // 2,)
            if kind == "up3d":
                sh[name + ".time_conv.weight"] = (2 * c, c, 3, 1, 1); sh[name + ".time_conv.bias"] = (2 * c,)
    c_out = dims[-1]
    sh["decoder.head.0.gamma"] = (c_out, 1, 1, 1)
    sh["decoder.head.2.weight"] = (3, c_out, 3, 3, 3); sh["decoder.head.2.bias"] = (3,)
    return sh


def synth_vae_state_dict(cfg: VaeConfig, seed: int = 0, device="cpu", dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """Random-init decoder weights (conv weights U(+-sqrt(3/fan_in)) so activations keep unit scale, gammas 1 + N(0,.1),
    biases N(0,.02); proj.weight non-zero, unlike the reference's zero init vae.py:237, so the attention block matters)."""
    sd = {}
    for name, shape in vae_decoder_param_shapes(cfg).items():
        if name.endswith("gamma"):
            w = 1.0 + 0.1 * hash_normal(seed, name, shape, device)
        elif name.endswith(".bias"):
            w = 0.02 * hash_normal(seed, name, shape, device)
        else:
            fan_in = int(math.prod(shape[1:]))
            w = (hash_uniform(seed, name, shape, device) * 2.0 - 1.0) * math.sqrt(3.0 / fan_in)
        sd[name] = w.to(dtype)
    return sd


# ---- umT5 text encoder ------------------------------------------------------------------
@dataclass
class T5Config:
    """umt5_xxl encoder (wan/modules/t5.py:466-479): per-layer relative position embedding (shared_pos=False)."""
    vocab_size: int = 256384
    dim: int = 4096
    dim_attn: int = 4096
    dim_ffn: int = 10240
    num_heads: int = 64
    num_layers: int = 24
    num_buckets: int = 32
    text_len: int = 512
    max_dist: int = 128


def t5_param_shapes(cfg: T5Config) -> Dict[str, Tuple[int, ...]]:
    """State-dict names of T5Encoder (wan/modules/t5.py:267-304)."""
    sh: Dict[str, Tuple[int, ...]] = {"token_embedding.weight": (cfg.vocab_size, cfg.dim)}
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        sh[p + "norm1.weight"] = (cfg.dim,)
        for n in "qkv":
            sh[p + f"attn.{n}.weight"] = (cfg.dim_attn, cfg.dim)
        sh[p + "attn.o.weight"] = (cfg.dim, cfg.dim_attn)
        sh[p + "norm2.weight"] = (cfg.dim,)
        sh[p + "ffn.gate.0.weight"] = (cfg.dim_ffn, cfg.dim)
        sh[p + "ffn.fc1.weight"] = (cfg.dim_ffn, cfg.dim)
        sh[p + "ffn.fc2.weight"] = (cfg.dim, cfg.dim_ffn)
        sh[p + "pos_embedding.embedding.weight"] = (cfg.num_buckets, cfg.num_heads)
    sh["norm.weight"] = (cfg.dim,)
    return sh


def synth_t5_state_dict(cfg: T5Config, seed: int = 0, device="cpu", dtype=torch.bfloat16, only: str = None) -> Dict[str, torch.Tensor]:
    """Random-init with the reference's init statistics (init_weights, t5.py:27-44); deviations so every term matters:
    norm weights 1 + N(0,.1), q weights 8x larger (the reference's (dim*dim_attn)^-.5 makes all logits ~0), position
    embeddings N(0, 1) (the reference's std .016 would make the relative bias invisible in bf16 parity)."""
    head_dim = cfg.dim_attn // cfg.num_heads
    sd = {}
    for name, shape in t5_param_shapes(cfg).items():
        if only is not None and name != only:           # one tensor of a model too large to hold twice (the 24-layer golden)
            continue
        z = hash_normal(seed, "t5." + name, shape, device)
        if name.endswith(("norm1.weight", "norm2.weight", "norm.weight")):
            w = 1.0 + 0.1 * z
        elif name == "token_embedding.weight":
            w = z
        elif name.endswith("attn.q.weight"):
            w = z * (cfg.dim ** -0.5) * (head_dim ** -0.25)
        elif name.endswith(("attn.k.weight", "attn.v.weight", "gate.0.weight", "fc1.weight")):
            w = z * cfg.dim ** -0.5
        elif name.endswith("attn.o.weight"):
            w = z * cfg.dim_attn ** -0.5
        elif name.endswith("fc2.weight"):
            w = z * cfg.dim_ffn ** -0.5
        else:                                   # pos_embedding
            w = z
        sd[name] = w.to(dtype)
    return sd


def synth_token_ids(cfg: T5Config, n_tokens: int, seed: int = 0, batch: int = 1):
    """(ids int64 [B, text_len] padded with 0, mask int64 [B, text_len]) as HuggingfaceTokenizer returns them
    (wan/modules/tokenizers.py:52-73: padding='max_length')."""
    u = hash_uniform(seed, "t5.ids", (batch, cfg.text_len))
    ids = (u * (cfg.vocab_size - 2)).long() + 2
    mask = torch.zeros(batch, cfg.text_len, dtype=torch.long)
    for b in range(batch):
        n = max(1, n_tokens - 3 * b)
        mask[b, :n] = 1
        ids[b, n - 1] = 1                      # </s>
        ids[b, n:] = 0                         # <pad>
    return ids, mask
