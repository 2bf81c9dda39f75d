"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (longlive_amd/).

CPU restatement of the Wan VAE *decoder* as LongLive uses it (wan/modules/vae.py): WanVAE_.decode / cached_decode
(:545-593) feed the latent frames one at a time through Decoder3d (:369-472) with per-convolution feature caches
(CACHE_T = 2, :14).  Functional over a state dict.  Pinned bit-exact to the reference's own classes by
tests/golden/vae_*.pt (oracle/make_golden.py::gen_vae) in tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
CACHE_T = 2


def rms_norm(x: Tensor, gamma: Tensor) -> Tensor:
    """RMS_norm.forward, channel_first (vae.py:51-55): F.normalize(x, dim=1) * sqrt(C) * gamma."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma


class RefVaeDecoder:
    def __init__(self, sd: Dict[str, Tensor], layers, dtype=torch.bfloat16):
        self.sd = {k: v.to(dtype) for k, v in sd.items()}
        self.layers = layers
        self.cache: List = []
        self.reset()

    def reset(self):
        """WanVAE_.clear_cache (:602-610)."""
        self.cache = [None] * 64
        self.idx = 0

    # CausalConv3d.forward (:28-36)
    def causal_conv(self, x: Tensor, name: str, cache_x: Optional[Tensor] = None, stride=(1, 1, 1)) -> Tensor:
        w, b = self.sd[name + ".weight"], self.sd[name + ".bias"]
        kt, kh, kw = w.shape[2:]
        pad = [kw // 2, kw // 2, kh // 2, kh // 2, 2 * (kt // 2), 0]
        if cache_x is not None and pad[4] > 0:
            x = torch.cat([cache_x, x], dim=2)
            pad[4] -= cache_x.shape[2]
        return F.conv3d(F.pad(x, pad), w, b)

    def _cached_conv(self, x: Tensor, name: str) -> Tensor:
        """The caching idiom repeated at every 3x3x3 conv (:205-218, :426-438, :455-470)."""
        i = self.idx
        cache_x = x[:, :, -CACHE_T:].clone()
        if cache_x.shape[2] < 2 and self.cache[i] is not None:
            cache_x = torch.cat([self.cache[i][:, :, -1:], cache_x], dim=2)
        y = self.causal_conv(x, name, self.cache[i])
        self.cache[i] = cache_x
        self.idx += 1
        return y

    def res_block(self, x: Tensor, name: str) -> Tensor:        # ResidualBlock.forward (:202-220)
        h = self.causal_conv(x, name + ".shortcut") if (name + ".shortcut.weight") in self.sd else x
        y = F.silu(rms_norm(x, self.sd[name + ".residual.0.gamma"]))
        y = self._cached_conv(y, name + ".residual.2")
        y = F.silu(rms_norm(y, self.sd[name + ".residual.3.gamma"]))
        y = self._cached_conv(y, name + ".residual.6")
        return y + h

    def attn_block(self, x: Tensor, name: str) -> Tensor:       # AttentionBlock.forward (:240-262)
        b, c, t, h, w = x.shape
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        y = rms_norm(y, self.sd[name + ".norm.gamma"])
        qkv = F.conv2d(y, self.sd[name + ".to_qkv.weight"], self.sd[name + ".to_qkv.bias"])
        q, k, v = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
        y = F.scaled_dot_product_attention(q, k, v)
        y = y.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
        y = F.conv2d(y, self.sd[name + ".proj.weight"], self.sd[name + ".proj.bias"])
        y = y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
        return y + x

    def resample(self, x: Tensor, name: str, mode: str) -> Tensor:   # Resample.forward, upsample modes (:101-143)
        b, c, t, h, w = x.shape
        if mode == "up3d":
            i = self.idx
            if self.cache[i] is None:
                self.cache[i] = "Rep"
                self.idx += 1
            else:
                cache_x = x[:, :, -CACHE_T:].clone()
                if cache_x.shape[2] < 2 and not isinstance(self.cache[i], str):
                    cache_x = torch.cat([self.cache[i][:, :, -1:], cache_x], dim=2)
                if cache_x.shape[2] < 2 and isinstance(self.cache[i], str):
                    cache_x = torch.cat([torch.zeros_like(cache_x), cache_x], dim=2)
                if isinstance(self.cache[i], str):
                    x = self.causal_conv(x, name + ".time_conv")
                else:
                    x = self.causal_conv(x, name + ".time_conv", self.cache[i])
                self.cache[i] = cache_x
                self.idx += 1
                x = x.reshape(b, 2, c, t, h, w)
                x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, w)
        t = x.shape[2]
        y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
        y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest").type_as(y)      # Upsample (:57-63)
        y = F.conv2d(y, self.sd[name + ".resample.1.weight"], self.sd[name + ".resample.1.bias"], padding=1)
        return y.reshape(b, t, c // 2, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)

    def decoder_step(self, x: Tensor) -> Tensor:                  # Decoder3d.forward with feat_cache (:423-472)
        self.idx = 0
        x = self._cached_conv(x, "decoder.conv1")
        for L in self.layers:
            if L[0] == "res":
                x = self.res_block(x, L[1])
            elif L[0] == "attn":
                x = self.attn_block(x, L[1])
            else:
                x = self.resample(x, L[1], L[0])
        x = F.silu(rms_norm(x, self.sd["decoder.head.0.gamma"]))
        return self._cached_conv(x, "decoder.head.2")

    def decode(self, z: Tensor, mean: Tensor, inv_std: Tensor, keep_cache: bool = False) -> Tensor:
        """WanVAE_.decode / cached_decode (:545-593).  z [1, 16, T, h, w] -> [1, 3, 1 + 4 (T-1), 8h, 8w] on a fresh cache."""
        if not keep_cache:
            self.reset()
        z = z / inv_std.view(1, -1, 1, 1, 1) + mean.view(1, -1, 1, 1, 1)
        x = F.conv3d(z, self.sd["conv2.weight"], self.sd["conv2.bias"])
        outs = [self.decoder_step(x[:, :, i:i + 1]) for i in range(z.shape[2])]
        if not keep_cache:
            self.reset()
        return torch.cat(outs, 2)


VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
            -0.1922, -0.9497, 0.2503, -0.2921]
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
           1.1253, 2.8251, 1.9160]


def decode_to_pixel(dec: RefVaeDecoder, latent: Tensor, use_cache: bool = False) -> Tensor:
    """WanVAEWrapper.decode_to_pixel (utils/wan_wrapper.py:96-117): latent [B, T, 16, h, w] -> [B, T', 3, H, W] fp32."""
    zs = latent.permute(0, 2, 1, 3, 4)
    mean = torch.tensor(VAE_MEAN, dtype=torch.float32).to(latent.dtype)
    inv_std = 1.0 / torch.tensor(VAE_STD, dtype=torch.float32).to(latent.dtype)     # wan_wrapper.py:102-103: bf16 division
    out = [dec.decode(u.unsqueeze(0), mean, inv_std, keep_cache=use_cache).float().clamp_(-1, 1).squeeze(0) for u in zs]
    return torch.stack(out, 0).permute(0, 2, 1, 3, 4)
