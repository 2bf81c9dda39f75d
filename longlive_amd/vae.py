"""Wan VAE decoder on the MI355X: the `vae` object the reference pipelines call once per
video (`self.vae.decode_to_pixel(output, use_cache=False)`, pipeline/causal_inference.py:249) or once per streamed
chunk (use_cache=True), behind the reference's WanVAEWrapper interface (utils/wan_wrapper.py:83-117) and the reference's
state-dict names (`decoder.*`, `conv2.*` of wan/modules/vae.py::WanVAE_).

Design (not a translation of the module tree): activations are channels-last bf16 [T, H, W, C]; every CausalConv3d /
Conv2d is ONE implicit-GEMM launch (ll_conv_cl) that gathers its shifted input pixels -- including the two cached
frames of temporal context, the zero padding and the nearest x2 upsample -- straight into LDS; RMS_norm + SiLU is one row
kernel; the middle attention block is GEMM + softmax + GEMM.  The reference feeds latent frames one at a time
(vae.py:555-569); causal convolutions make any chunking equivalent, so after the first frame (whose temporal upsampling
is skipped, the 'Rep' branch of Resample.forward, vae.py:108-112) frames are processed `chunk` at a time to fill the
256 CUs.  There is no CPU path: without liblonglive_hip.so every call raises.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .synth import VaeConfig, vae_decoder_layout, vae_decoder_param_shapes

bf16 = torch.bfloat16

VAE_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
            -0.1922, -0.9497, 0.2503, -0.2921]
VAE_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
           1.1253, 2.8251, 1.9160]


class _Conv:
    """One convolution: packed weights + (for temporal kernels) its input buffer [2 + T, H, W, Cin], whose first two frames
    are the stream's previous two input frames -- the reference's feat_cache (vae.py:29-34, 207-216) -- kept in the layout
    the kernel reads, so the history costs one 2-frame copy per call and no concatenation."""

    def __init__(self, w: torch.Tensor, b: torch.Tensor):
        self.w, self.b, self.geo = ops.pack_conv_weight(w, b)
        self.temporal = self.geo[3] > 1
        self._hist: Optional[torch.Tensor] = None       # view of the last two input frames seen so far
        self._buf: Optional[torch.Tensor] = None

    def reset(self):
        self._hist = None
        self._buf = None

    def input(self, T: int, H: int, W: int, device) -> torch.Tensor:
        """Where the producer should write this convolution's T input frames."""
        cin = self.geo[0]
        if not self.temporal:
            return torch.empty(T, H, W, cin, dtype=bf16, device=device)
        buf = torch.empty(2 + T, H, W, cin, dtype=bf16, device=device)
        if self._hist is None:
            buf[:2].zero_()
        else:
            buf[:2].copy_(self._hist)
        self._buf = buf
        return buf[2:]

    def __call__(self, x: torch.Tensor, upsample: bool = False, res: Optional[torch.Tensor] = None, rms=None,
                 want_raw: bool = True) -> torch.Tensor:
        """rms = (gamma, out_rms): the convolution's epilogue also writes rms_silu(result) into out_rms (the next convolution's input
        buffer) -- one launch instead of two where ops.conv_cl_rms_ok says so, else the two launches; want_raw = False: the
        un-normalised result is not needed (None is returned when the fused kernel ran)."""
        if not self.temporal:
            buf = x
        else:
            if self._buf is None or x.data_ptr() != self._buf[2:].data_ptr():
                self.input(x.shape[0], x.shape[1], x.shape[2], x.device).copy_(x)     # producer did not write in place
            buf, self._buf = self._buf, None
        if rms is not None and ops.conv_cl_rms_ok(self.geo, x.shape[1], x.shape[2], upsample):
            y = ops.conv_cl_rms(buf, self.w, self.b, self.geo, rms[0], rms[1], upsample=upsample, res=res, want_raw=want_raw)
        else:
            y = ops.conv_cl(buf, self.w, self.b, self.geo, upsample=upsample, res=res)
            if rms is not None:
                ops.rms_silu_cl(y, rms[0], out=rms[1])
        if self.temporal:
            self._hist = buf[-2:]
        return y


class WanVAEDecoderHIP(nn.Module):
    """Decoder3d + conv2 of WanVAE_ (vae.py:369-472, 545-593).  Parameters carry the reference's names so
    `load_state_dict(torch.load('Wan2.1_VAE.pth'), strict=False)` works (encoder keys are ignored)."""

    def __init__(self, cfg: Optional[VaeConfig] = None, device="cuda", chunk: int = 2):
        super().__init__()
        self.fuse_rms = os.environ.get("LL_VAE_FUSE", "1") != "0"      # kernel A/B only: 0 = RMS_norm + SiLU as their own launches
        self.cfg = cfg or VaeConfig()
        self.dims, self.layers = vae_decoder_layout(self.cfg)
        self.chunk = max(1, int(chunk))
        self._names: Dict[str, str] = {}
        for name, shape in vae_decoder_param_shapes(self.cfg).items():
            reg = name.replace(".", "__")
            self._names[name] = reg
            self.register_parameter(reg, nn.Parameter(torch.zeros(shape, dtype=bf16, device=device), requires_grad=False))
        self._packed = False
        self._convs: Dict[str, _Conv] = {}
        self._first = True

    # -- state dict with the reference's dotted names ---------------------------------------------------------------
    def state_dict(self, *a, **k):
        return {name: getattr(self, reg).data for name, reg in self._names.items()}

    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        missing = [n for n in self._names if n not in sd]
        unexpected = [n for n in sd if n not in self._names]
        if strict and (missing or [u for u in unexpected if not u.startswith(("encoder.", "conv1."))]):
            raise RuntimeError(f"WanVAEDecoderHIP.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for name, reg in self._names.items():
            if name in sd:
                p = getattr(self, reg)
                if tuple(sd[name].shape) != tuple(p.shape):
                    raise RuntimeError(f"{name}: shape {tuple(sd[name].shape)} != {tuple(p.shape)}")
                p.data.copy_(sd[name].to(device=p.device, dtype=bf16))
        self._packed = False
        return missing, unexpected

    def _p(self, name: str) -> torch.Tensor:
        return getattr(self, self._names[name]).data

    def _pack(self):
        """One-time re-layout of the weights for the kernels."""
        c = {}
        names = [n[:-len(".weight")] for n in self._names if n.endswith(".weight")]
        for n in names:
            if ".to_qkv" in n or ".proj" in n:
                continue
            c[n] = _Conv(self._p(n + ".weight"), self._p(n + ".bias"))
        self._convs = c
        self._attn = {}
        for L in self.layers:
            if L[0] == "attn":
                name, ch = L[1], L[2]
                wqkv = self._p(name + ".to_qkv.weight").reshape(3 * ch, ch)
                bqkv = self._p(name + ".to_qkv.bias")
                self._attn[name] = dict(
                    w=[wqkv[i * ch:(i + 1) * ch].contiguous() for i in range(3)],
                    b=[bqkv[i * ch:(i + 1) * ch].contiguous() for i in range(3)],
                    wo=self._p(name + ".proj.weight").reshape(ch, ch).contiguous(), bo=self._p(name + ".proj.bias").contiguous(),
                    gamma=self._p(name + ".norm.gamma").reshape(-1).contiguous())
        self._gamma = {n: self._p(n).reshape(-1).contiguous() for n in self._names if n.endswith("gamma")}
        self._mean = torch.tensor(VAE_MEAN[:self.cfg.z_dim], dtype=torch.float32).to(bf16).to(self._p("conv2.bias").device)
        std = torch.tensor(VAE_STD[:self.cfg.z_dim], dtype=torch.float32).to(bf16)
        self._inv_std = (1.0 / std).to(self._mean.device)                # utils/wan_wrapper.py:102-103 (bf16 division)
        self._packed = True

    # -- streaming state -------------------------------------------------------------------------------------------
    def clear_cache(self):
        """WanVAE_.clear_cache (vae.py:602-610)."""
        for c in self._convs.values():
            c.reset()
        self._first = True

    # -- blocks ----------------------------------------------------------------------------------------------------
    def _res_block(self, x, name, pre: bool = False, nxt=None):           # ResidualBlock.forward (vae.py:202-220)
        """pre: the producer of x already wrote rms_silu(x) into this block's first convolution's input buffer.  nxt = (gamma, conv) of
        the RMS_norm + SiLU + convolution that consume this block's output next: written by conv2's epilogue (one launch where
        ops.conv_cl_rms_ok, else conv + rms_silu) -- the un-normalised output is returned as well (the next block's shortcut)."""
        T, H, W, _ = x.shape
        c1, c2 = self._convs[name + ".residual.2"], self._convs[name + ".residual.6"]
        h = self._convs[name + ".shortcut"](x) if (name + ".shortcut") in self._convs else x
        if pre:
            y = c1._buf[2:] if c1.temporal else c1._buf
        else:
            y = ops.rms_silu_cl(x, self._gamma[name + ".residual.0.gamma"], out=c1.input(T, H, W, x.device))
        y2 = c2.input(T, H, W, x.device)
        if self.fuse_rms:
            c1(y, rms=(self._gamma[name + ".residual.3.gamma"], y2), want_raw=False)  # conv -> RMS_norm -> SiLU in one launch where covered
        else:
            ops.rms_silu_cl(c1(y), self._gamma[name + ".residual.3.gamma"], out=y2)
        if nxt is None:
            return c2(y2, res=h)
        return c2(y2, res=h, rms=(nxt[0], nxt[1].input(T, H, W, x.device)), want_raw=True)

    def _attn_block(self, x, name):                                      # AttentionBlock.forward (vae.py:240-262)
        a = self._attn[name]
        T, H, W, C = x.shape
        hw = H * W
        hwp = (hw + 63) // 64 * 64
        out = torch.empty_like(x)
        zeros_hw = torch.zeros(hwp, dtype=bf16, device=x.device)
        zeros_c = torch.zeros(C, dtype=bf16, device=x.device)
        for t in range(T):
            xt = x[t].reshape(hw, C)
            y = ops.rms_silu_cl(xt, a["gamma"], silu=False)
            q = ops.gemm(y, a["w"][0], a["b"][0])
            k = torch.zeros(hwp, C, dtype=bf16, device=x.device)
            ops.gemm(y, a["w"][1], a["b"][1], out=k[:hw])
            v = ops.gemm(y, a["w"][2], a["b"][2])
            vt = torch.zeros(C, hwp, dtype=bf16, device=x.device)
            vt[:, :hw].copy_(v.t())
            s = ops.gemm(q, k, zeros_hw)                                 # [hw, hwp] = q k^T
            p = ops.softmax_rows(s, 1.0 / math.sqrt(C), n_valid=hw)
            o = ops.gemm(p, vt, zeros_c)                                 # [hw, C] = p v
            ops.gemm(o, a["wo"], a["bo"], epilogue=ops.EPI_BIAS_RES, res=xt, out=out[t].reshape(hw, C))
        return out

    def _resample(self, x, name, mode, nxt=None):                                  # Resample.forward (vae.py:101-143)
        if mode == "up3d" and not self._first:
            T, H, W, C = x.shape
            y = self._convs[name + ".time_conv"](x)                      # [T,H,W,2C]: two output frames per input frame
            x = y.view(T, H, W, 2, C).permute(0, 3, 1, 2, 4).reshape(2 * T, H, W, C).contiguous()   # vae.py:131-134 interleave
        conv = self._convs[name + ".resample.1"]
        if nxt is None:
            return conv(x, upsample=True)
        return conv(x, upsample=True, rms=(nxt[0], nxt[1].input(x.shape[0], 2 * x.shape[1], 2 * x.shape[2], x.device)), want_raw=True)

    def _consumer(self, i: int):
        """(gamma, conv) of the RMS_norm + SiLU + convolution that read layer i's output next -- a residual block's first pair, or the
        head's -- when the producer can write the normalised tensor itself; None: an attention block or a resample comes next."""
        if not self.fuse_rms:
            return None
        if i + 1 == len(self.layers):
            c = (self._gamma["decoder.head.0.gamma"], self._convs["decoder.head.2"])
        elif self.layers[i + 1][0] == "res":
            nl = self.layers[i + 1]
            c = (self._gamma[nl[1] + ".residual.0.gamma"], self._convs[nl[1] + ".residual.2"])
        else:
            return None
        return c if c[1].temporal else None           # (the consumer's input buffer is remembered only by temporal convolutions)

    def _decoder_step(self, x):                                          # Decoder3d.forward (vae.py:423-472)
        x = self._convs["decoder.conv1"](x)
        pre = False
        for i, L in enumerate(self.layers):
            nxt = self._consumer(i)
            if L[0] == "res":
                x = self._res_block(x, L[1], pre, nxt)
            elif L[0] == "attn":
                x, nxt = self._attn_block(x, L[1]), None
            else:
                x = self._resample(x, L[1], L[0], nxt)
            pre = nxt is not None
        head = self._convs["decoder.head.2"]
        if pre:
            y = head._buf[2:] if head.temporal else head._buf
        else:
            y = ops.rms_silu_cl(x, self._gamma["decoder.head.0.gamma"], out=head.input(x.shape[0], x.shape[1], x.shape[2], x.device))
        return head(y)                          # [T', H, W, 8] (3 channels + padding)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, keep_cache: bool = False) -> torch.Tensor:
        """z [T, 16, h, w] bf16 latent frames of one video -> fp32 [T', 3, 8h, 8w] in [-1, 1];
        T' = 1 + 4 (T - 1) on a fresh cache, 4 T when continuing a stream (WanVAE_.decode / cached_decode, vae.py:545-593)."""
        if not self._packed:
            self._pack()
        if not keep_cache:
            self.clear_cache()
        x = ops.vae_unscale_cl(z.contiguous(), self._mean, self._inv_std)
        x = self._convs["conv2"](x)
        outs: List[torch.Tensor] = []
        i, T = 0, x.shape[0]
        while i < T:
            n = 1 if self._first else min(self.chunk, T - i)
            y = self._decoder_step(x[i:i + n])
            outs.append(ops.cl_to_tchw_clamp(y))
            self._first = False
            i += n
        if not keep_cache:
            self.clear_cache()
        return torch.cat(outs, 0)


class WanVAEWrapper(nn.Module):
    """Drop-in for utils/wan_wrapper.py::WanVAEWrapper on the decode side (the only side the inference path uses,
    SURVEY.md section 8f): `decode_to_pixel(latent [B,T,16,h,w], use_cache) -> [B,T',3,H,W]` fp32 in [-1,1]."""

    def __init__(self, cfg: Optional[VaeConfig] = None, device="cuda", chunk: int = 2):
        super().__init__()
        self.model = WanVAEDecoderHIP(cfg, device=device, chunk=chunk)
        self.mean = torch.tensor(VAE_MEAN, dtype=torch.float32)
        self.std = torch.tensor(VAE_STD, dtype=torch.float32)

    def load_state_dict(self, sd, strict: bool = True, assign: bool = False):
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()}
        return self.model.load_state_dict(sd, strict=strict)

    def decode_to_pixel(self, latent: torch.Tensor, use_cache: bool = False) -> torch.Tensor:
        if latent.dtype != bf16:
            latent = latent.to(bf16)
        if use_cache:      # ONE streaming feature cache: a second sample would continue the first one's stream
            assert latent.shape[0] == 1, "Batch size must be 1 when using cache"          # utils/wan_wrapper.py:99
        out = [self.model.decode(u, keep_cache=use_cache) for u in latent]
        return torch.stack(out, 0)

    def encode_to_latent(self, pixel):
        raise NotImplementedError("longlive_amd: the VAE encoder is outside the inference hot path (SURVEY.md section 8)")
