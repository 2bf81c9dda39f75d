"""Tensor-level wrappers over the C ABI.  torch is plumbing here (device memory + the current HIP stream);
all arithmetic happens in liblonglive_hip.so.  Every wrapper checks operand shapes on the host before a launch:
a kernel that faults can take the whole 8-GPU host down.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import _lib

bf16 = torch.bfloat16

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_GATE_RES, EPI_BIAS_RES = 0, 1, 2, 3


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event timing of individual launches on the stream they are launched on (bench.py's roofline leg).
    `ops.timer = KernelTimer()` turns it on; wrappers call timer.around(tag, work) for tagged kernels."""

    def __init__(self, tags=None):
        self.tags = None if tags is None else set(tags)
        self.records = {}          # tag -> [(start_event, end_event, work)]
        self.base = torch.cuda.Event(enable_timing=True)      # origin of the wall-clock intervals of union_ms()
        self.base.record()

    def begin(self, tag):
        if self.tags is not None and tag not in self.tags:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, tag, start, work):
        if start is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.setdefault(tag, []).append((start, ev, work))

    def summary(self):
        """tag -> dict(launches, avg_ms, total_ms, work_per_launch) (call after a device sync)."""
        out = {}
        for tag, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[tag] = dict(launches=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                            work_per_launch=sum(w for _, _, w in recs) / len(recs))
        return out


    def union_ms(self, tags) -> float:
        """Wall time during which at least one launch of `tags` was in flight (launches on two HIP streams overlap: the sum of
        their durations counts the shared time twice).  Call after a device sync."""
        iv = sorted((self.base.elapsed_time(a), self.base.elapsed_time(b)) for t in tags for a, b, _ in self.records.get(t, []))
        total, cur0, cur1 = 0.0, None, None
        for a, b in iv:
            if cur1 is None or a > cur1:
                if cur1 is not None:
                    total += cur1 - cur0
                cur0, cur1 = a, b
            else:
                cur1 = max(cur1, b)
        return total + ((cur1 - cur0) if cur1 is not None else 0.0)


timer: Optional[KernelTimer] = None


def _t0(tag: str):
    return timer.begin(tag) if timer is not None else None


def _t1(tag: str, ev, work: float):
    """work = the launch's ALGORITHMIC FLOPs (MFMA kernels) or bytes (HBM-bound row kernels), SURVEY.md section 8d."""
    if timer is not None:
        timer.end(tag, ev, work)


def _chk(t: torch.Tensor, name: str, dtype=bf16) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a device tensor (longlive_amd has no CPU path)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor, got strides {t.stride()}")
    return t


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def ln_modulate(x, e, mod, shift_idx: int, scale_idx: int, num_frames: int, eps: float, out=None, tag: str = "ln_modulate"):
    """x [B,L,C]; e [B,F,nmod,C]; mod [nmod,C] (causal_model.py:445,463-464,506-507); mod=None: `e` is a layer's slice of
    modulation_table(), i.e. already bf16(mod + e)."""
    _chk(x, "x"); _chk(e, "e")
    B, L, Cc = x.shape
    nmod = e.shape[-2]
    assert e.shape == (B, num_frames, nmod, Cc), (e.shape, (B, num_frames, nmod, Cc))
    if mod is not None:
        _chk(mod, "mod")
        assert mod.numel() == nmod * Cc
    out = torch.empty_like(x) if out is None else _chk(out, "out")
    assert out.shape == x.shape
    lib = _lib.load()
    t0 = _t0(tag)
    _lib.check(lib.ll_ln_modulate(x.data_ptr(), out.data_ptr(), e.data_ptr(), _ptr(mod), nmod, shift_idx,
                                  scale_idx, B, L, Cc, num_frames, eps, _stream()), "ll_ln_modulate")
    _t1(tag, t0, 4.0 * x.numel())                    # read x, write out (bf16)
    return out


def ln_modulate_q8(x, e, mod, shift_idx: int, scale_idx: int, num_frames: int, eps: float, tag: str = "ln_modulate"):
    """ln_modulate emitting (int8 [B,L,C], float32 scale [B*L]) for a following W8A8 GEMM."""
    _chk(x, "x"); _chk(e, "e")
    B, L, Cc = x.shape
    nmod = e.shape[-2]
    assert e.shape == (B, num_frames, nmod, Cc)
    if mod is not None:
        _chk(mod, "mod")
        assert mod.numel() == nmod * Cc
    q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    sc = torch.empty(B * L, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    t0 = _t0(tag)
    _lib.check(lib.ll_ln_modulate_q8(x.data_ptr(), q.data_ptr(), sc.data_ptr(), e.data_ptr(), _ptr(mod), nmod,
                                     shift_idx, scale_idx, B, L, Cc, num_frames, eps, _stream()), "ll_ln_modulate_q8")
    _t1(tag, t0, 3.0 * x.numel())                    # read bf16, write int8
    return q, sc


def modulation_table_f32(e, mods, one_plus_mask: int):
    """The fp32 form for ln_modulate_tab: [NL,B,F,nmod,C] float32 = float(bf16(mods[l] + e)), chunks in one_plus_mask
    float(bf16(1 + bf16(mods[l] + e))) -- the `1 + e[1]` of causal_model.py:445,463 once per (layer, frame)."""
    _chk(e, "e"); _chk(mods, "mods")
    B, F, nmod, Cc = e.shape
    NL = mods.shape[0]
    assert mods.shape == (NL, nmod, Cc), (mods.shape, e.shape)
    out = torch.empty(NL, B, F, nmod, Cc, dtype=torch.float32, device=e.device)
    lib = _lib.load()
    _lib.check(lib.ll_modulation_table_f32(e.data_ptr(), mods.data_ptr(), out.data_ptr(), NL, B * F, nmod, Cc, int(one_plus_mask),
                                           _stream()), "ll_modulation_table_f32")
    return out


def ln_modulate_tab(x, tab, shift_idx: int, scale_idx: int, num_frames: int, eps: float, q8: bool = False, tag: str = "ln_modulate"):
    """ln_modulate / ln_modulate_q8 from a layer's slice tab [B,F,nmod,C] (float32) of modulation_table_f32, whose scale chunk
    already holds 1 + scale: same bits.  Returns bf16 [B,L,C], or (int8 [B,L,C], float32 scale [B*L]) when q8."""
    _chk(x, "x"); _chk(tab, "tab", torch.float32)
    B, L, Cc = x.shape
    nmod = tab.shape[2]
    assert tab.shape == (B, num_frames, nmod, Cc), (tab.shape, (B, num_frames, nmod, Cc))
    lib = _lib.load()
    if q8:
        q = torch.empty(B, L, Cc, dtype=torch.int8, device=x.device)
        sc = torch.empty(B * L, dtype=torch.float32, device=x.device)
        out = None
    else:
        out = torch.empty_like(x)
        q = sc = None
    t0 = _t0(tag)
    _lib.check(lib.ll_ln_modulate_tab(x.data_ptr(), _ptr(out), _ptr(q), _ptr(sc), tab.data_ptr(), nmod, shift_idx, scale_idx,
                                      B, L, Cc, num_frames, eps, _stream()), "ll_ln_modulate_tab")
    _t1(tag, t0, (3.0 if q8 else 4.0) * B * L * Cc)
    return (q, sc) if q8 else out


def modulation_table(e, mods):
    """e [B,F,nmod,C] (per forward), mods [NL,nmod,C] (per layer) -> [NL,B,F,nmod,C] = bf16(mods[l] + e): what every block
    computes as `self.modulation.unsqueeze(1) + e` (causal_model.py:440), for all layers in one launch."""
    _chk(e, "e"); _chk(mods, "mods")
    B, F, nmod, Cc = e.shape
    NL = mods.shape[0]
    assert mods.shape == (NL, nmod, Cc), (mods.shape, e.shape)
    out = torch.empty(NL, B, F, nmod, Cc, dtype=bf16, device=e.device)
    lib = _lib.load()
    _lib.check(lib.ll_modulation_table(e.data_ptr(), mods.data_ptr(), out.data_ptr(), NL, B * F, nmod, Cc, _stream()),
               "ll_modulation_table")
    return out


def layernorm_affine_q8(x, w, b, eps: float):
    _chk(x, "x"); _chk(w, "w"); _chk(b, "b")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    sc = torch.empty(rows, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    t0 = _t0("layernorm_affine")
    _lib.check(lib.ll_layernorm_affine_q8(x.data_ptr(), w.data_ptr(), b.data_ptr(), q.data_ptr(), sc.data_ptr(), rows, Cc,
                                          eps, _stream()), "ll_layernorm_affine_q8")
    _t1("layernorm_affine", t0, 3.0 * x.numel())
    return q, sc


def layernorm_affine(x, w, b, eps: float, out=None):
    _chk(x, "x"); _chk(w, "w"); _chk(b, "b")
    Cc = x.shape[-1]
    assert w.numel() == Cc and b.numel() == Cc
    out = torch.empty_like(x) if out is None else _chk(out, "out")
    assert out.shape == x.shape
    lib = _lib.load()
    t0 = _t0("layernorm_affine")
    _lib.check(lib.ll_layernorm_affine(x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), x.numel() // Cc, Cc,
                                       eps, _stream()), "ll_layernorm_affine")
    _t1("layernorm_affine", t0, 4.0 * x.numel())
    return out


def rmsnorm(x, w, eps: float, out=None, C: Optional[int] = None):
    """RMSNorm over the first C columns of each row of a 2-D (or flattened) row-major view.  x may be a column
    slice of a wider buffer as long as its last-dim stride is 1 and rows are evenly strided."""
    assert x.dtype == bf16 and x.is_cuda and x.stride(-1) == 1
    Cc = x.shape[-1] if C is None else C
    x2 = x.flatten(0, -2) if x.dim() > 2 else x
    assert x2.dim() == 2 and x2.data_ptr() == x.data_ptr(), "rows of x must be evenly strided"
    rows, ldx = x2.shape[0], x2.stride(0)
    _chk(w, "w")
    assert w.numel() == Cc
    if out is None:
        out = torch.empty(rows, Cc, dtype=bf16, device=x.device)
    assert out.dtype == bf16 and out.is_cuda and out.stride(-1) == 1
    o2 = out.flatten(0, -2) if out.dim() > 2 else out
    assert o2.dim() == 2 and o2.data_ptr() == out.data_ptr() and o2.shape[0] == rows and o2.shape[1] >= Cc
    lib = _lib.load()
    t0 = _t0("rmsnorm")
    _lib.check(lib.ll_rmsnorm(x2.data_ptr(), w.data_ptr(), o2.data_ptr(), rows, Cc, ldx, o2.stride(0), eps, _stream()),
               "ll_rmsnorm")
    _t1("rmsnorm", t0, 4.0 * rows * Cc)
    return out


def qk_norm_rope_kv_store(qkv, wq, wk, rope_f, rope_hw, q_out, cache_k, cache_v, head_dim: int, frame_len: int,
                          start_frame: int, write_start: int, roped_offset: int, write_len: int, eps: float):
    """qkv [B,L,3C] -> q_out [B,L,C]; cache_k/v [B,S,H,D] rows [write_start, +write_len) updated in place.  cache_v=None: V
    was inserted by gemm_qkv_v_insert and the v third of qkv is not read."""
    _chk(qkv, "qkv"); _chk(wq, "wq"); _chk(wk, "wk"); _chk(q_out, "q_out"); _chk(cache_k, "cache_k")
    if cache_v is not None:
        _chk(cache_v, "cache_v")
    _chk(rope_f, "rope_f", torch.float32); _chk(rope_hw, "rope_hw", torch.float32)
    B, L, C3 = qkv.shape
    Cc = C3 // 3
    assert C3 == 3 * Cc and q_out.shape == (B, L, Cc)
    S = cache_k.shape[1]
    assert cache_k.shape[0] == B and cache_k.numel() == B * S * Cc and (cache_v is None or cache_v.shape == cache_k.shape)
    half = head_dim // 2
    nf = half - 2 * (half // 3)
    assert rope_f.shape == (1024, nf, 2), rope_f.shape
    assert rope_hw.shape == (frame_len, half - nf, 2), rope_hw.shape
    lib = _lib.load()
    t0 = _t0("qk_norm_rope_kv_store")
    _lib.check(lib.ll_qk_norm_rope_kv_store(qkv.data_ptr(), wq.data_ptr(), wk.data_ptr(), rope_f.data_ptr(),
                                            rope_hw.data_ptr(), q_out.data_ptr(), cache_k.data_ptr(),
                                            _ptr(cache_v), B, L, Cc, head_dim, frame_len, start_frame, S,
                                            write_start, roped_offset, write_len, eps, _stream()),
               "ll_qk_norm_rope_kv_store")
    nv = 1 if cache_v is not None else 0
    _t1("qk_norm_rope_kv_store", t0, 2.0 * ((2 + nv) * B * L * Cc + B * L * Cc + (1 + nv) * B * write_len * Cc))   # q,k(,v) in; q out; k(,v) -> cache
    return q_out


def kv_roll(cache_k, cache_v, dst: int, src: int, n: int):
    _chk(cache_k, "cache_k"); _chk(cache_v, "cache_v")
    B, S = cache_k.shape[:2]
    Cc = cache_k.numel() // (B * S)
    assert cache_v.shape == cache_k.shape
    lib = _lib.load()
    t0 = _t0("kv_roll")
    _lib.check(lib.ll_kv_roll(cache_k.data_ptr(), cache_v.data_ptr(), B, S, Cc, dst, src, n, _stream()), "ll_kv_roll")
    _t1("kv_roll", t0, 2.0 * 2 * 2 * B * n * Cc)     # k and v, read + write, bf16


def gemm(x, w, bias, epilogue: int = EPI_BIAS, out=None, res=None, e=None, mod=None, gate_idx: int = 0,
         rows_per_batch: int = 0, frame_len: int = 0, tag: str = "gemm"):
    """out[M,N] = epilogue(x[M,K] @ w[N,K]^T + bias).  x may be any [..., K] contiguous tensor."""
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape == (N, K) and bias.numel() == N, (w.shape, bias.shape, K)
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=bf16, device=x.device)
    _chk(out, "out")
    assert out.numel() == M * N
    nmod = 0
    if epilogue in (EPI_BIAS_GATE_RES, EPI_BIAS_RES):
        _chk(res, "res")
        assert res.numel() == M * N
    if epilogue == EPI_BIAS_GATE_RES:
        _chk(e, "e")
        nmod = e.shape[-2]
        assert e.shape[-1] == N and e.numel() == (M // frame_len) * nmod * N, (e.shape, M, frame_len)
        if mod is not None:         # None: e already holds bf16(mod + e) (modulation_table)
            _chk(mod, "mod")
            assert mod.numel() == nmod * N
    lib = _lib.load()
    t0 = _t0(tag)
    if M <= 1024 and M * N <= (1 << 22) and epilogue in (EPI_BIAS, EPI_BIAS_RES) and _ksplit_plan(lib, M, N, K):
        # few rows (the text side: umT5 at 512 tokens, text K / V projections): K cut into ranges so that the grid fills the device.
        # Taken by itself only up to 1024 rows: the DiT's own linears (a 1-frame chunk has M = 1560) never come here, so the order
        # of a row's fp32 sum there does not depend on M or on the device's CU count.  Inside this region the number of ranges
        # depends on N and K only (gemm_ksplit_splits); WHETHER the path is taken still depends on M and the CU count.
        ws = ksplit_workspace(x.device, M, N, K)
        _lib.check(lib.ll_gemm_bf16_ksplit(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, K, N, epilogue,
                                           _ptr(res), ws.data_ptr(), ws.numel(), _stream()), "ll_gemm_bf16_ksplit")
    else:
        _lib.check(lib.ll_gemm_bf16(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, K, N, epilogue,
                                    _ptr(res), _ptr(e), _ptr(mod), nmod, gate_idx, rows_per_batch, frame_len, _stream()),
                   "ll_gemm_bf16")
    _t1(tag, t0, 2.0 * M * N * K)
    return out


def gemm_ssq_planes(M: int, N: int, K: int) -> int:
    """128-column planes ll_gemm_bf16_ssq would write for this shape under the current tuning (0 = not covered)."""
    return int(_lib.load().ll_gemm_ssq_planes(M, N, K))


def gemm_ssq(x, w, bias, out=None, tag: str = "gemm"):
    """(x @ w^T + bias [.., N] bf16, ssq [N / 128, M] fp32): the bias GEMM whose epilogue also leaves, per row and 128-column n-tile,
    the sum of squares of its bf16 outputs -- the statistics of the RMSNorm that follows a q projection (model.py:172), consumed by
    flash_attn_qnorm.  Only for shapes gemm_ssq_planes() accepts."""
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape == (N, K) and bias.numel() == N
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=bf16, device=x.device)
    _chk(out, "out")
    assert out.numel() == M * N
    planes = gemm_ssq_planes(M, N, K)
    if planes == 0:
        raise RuntimeError(f"gemm_ssq: {M} x {N} x {K} is not covered by the generated kernel (use gemm + rmsnorm)")
    ssq = torch.empty(planes, M, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    t0 = _t0(tag)
    _lib.check(lib.ll_gemm_bf16_ssq(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), ssq.data_ptr(), M, N, K, K, N, _stream()),
               "ll_gemm_bf16_ssq")
    _t1(tag, t0, 2.0 * M * N * K)
    return out, ssq


_ksplit_ws = {}


def _ksplit_plan(lib, M: int, N: int, K: int) -> int:
    """K-ranges the small-M path would use (0 = not taken).  Asked per call: the answer follows the library's tuning, and only
    calls with few rows get here (the DiT's block linears do not)."""
    return int(lib.ll_gemm_ksplit_plan(M, N, K))


def _dev_index(device) -> int:
    return torch.cuda.current_device() if device.index is None else device.index


def ksplit_workspace(device, M: int, N: int, K: int) -> torch.Tensor:
    """Scratch for ll_gemm_bf16_ksplit ([splits][M][N] fp32 tile sums; needs no initialisation), one buffer per (device, stream)."""
    need = int(_lib.load().ll_gemm_ksplit_workspace_bytes(M, N, K))
    key = (device.type, _dev_index(device), int(_stream() or 0))
    buf = _ksplit_ws.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(max(need, 16), dtype=torch.uint8, device=device)
        _ksplit_ws[key] = buf
    return buf


def gemm_qkv_v_insert(x, w, bias, cache_v, write_start: int, roped_offset: int, write_len: int, xq=None, tag: str = "gemm_qkv"):
    """The fused q|k|v projection x [B,L,K] @ w [3C,K]^T + bias with the V third inserted into cache_v [B,S,H,D] by the GEMM
    epilogue (token t -> slot write_start + t - roped_offset for 0 <= t - roped_offset < write_len).  Returns the [B,L,3C]
    buffer whose q and k thirds are valid (the v third is unwritten).  xq = (int8 [B,L,K], scale [B*L]) with int8 w = (wq, sw):
    the W8A8 form."""
    _chk(bias, "bias"); _chk(cache_v, "cache_v")
    int8 = xq is not None
    if int8:
        xi, sx = xq
        wq, sw = w
        _chk(xi, "xq", torch.int8); _chk(sx, "sx", torch.float32); _chk(wq, "wq", torch.int8); _chk(sw, "sw", torch.float32)
        B, L, K = xi.shape
        N = wq.shape[0]
        assert wq.shape == (N, K) and sx.numel() == B * L and sw.numel() == N
        dev = xi.device
    else:
        _chk(x, "x"); _chk(w, "w")
        B, L, K = x.shape
        N = w.shape[0]
        assert w.shape == (N, K)
        dev = x.device
    assert bias.numel() == N and N % 3 == 0
    S = cache_v.shape[1]
    assert cache_v.shape[0] == B and cache_v.numel() == B * S * (N // 3), (cache_v.shape, B, S, N)
    out = torch.empty(B, L, N, dtype=bf16, device=dev)
    lib = _lib.load()
    t0 = _t0(tag)
    if int8:
        _lib.check(lib.ll_gemm_w8a8_qkv(xi.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                        B * L, N, K, N, cache_v.data_ptr(), B, L, S, write_start, roped_offset, write_len, _stream()),
                   "ll_gemm_w8a8_qkv")
    else:
        _lib.check(lib.ll_gemm_bf16_qkv(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B * L, N, K, K, N,
                                        cache_v.data_ptr(), B, L, S, write_start, roped_offset, write_len, _stream()),
                   "ll_gemm_bf16_qkv")
    _t1(tag, t0, 2.0 * B * L * N * K)
    return out


def quantize_rows(x, q=None, scale=None):
    """Per-row symmetric int8 quantisation of a [..., K] bf16 tensor -> (int8 [..., K], float32 scale [rows])."""
    _chk(x, "x")
    K = x.shape[-1]
    rows = x.numel() // K
    if q is None:
        q = torch.empty(x.shape, dtype=torch.int8, device=x.device)
    if scale is None:
        scale = torch.empty(rows, dtype=torch.float32, device=x.device)
    _chk(q, "q", torch.int8); _chk(scale, "scale", torch.float32)
    assert q.numel() == x.numel() and scale.numel() == rows
    lib = _lib.load()
    t0 = _t0("quantize_rows")
    _lib.check(lib.ll_quantize_rows(x.data_ptr(), q.data_ptr(), scale.data_ptr(), rows, K, K, _stream()), "ll_quantize_rows")
    _t1("quantize_rows", t0, 3.0 * x.numel())
    return q, scale


def gemm_w8a8(xq, sx, wq, sw, bias, epilogue: int = EPI_BIAS, out=None, res=None, e=None, mod=None, gate_idx: int = 0,
              rows_per_batch: int = 0, frame_len: int = 0, tag: str = "gemm"):
    """out[M,N] = epilogue(sx[m] sw[n] (xq[M,K] @ wq[N,K]^T) + bias): int8 operands, int32 accumulation, bf16 out."""
    _chk(xq, "xq", torch.int8); _chk(wq, "wq", torch.int8); _chk(sx, "sx", torch.float32); _chk(sw, "sw", torch.float32)
    _chk(bias, "bias")
    K = xq.shape[-1]
    M = xq.numel() // K
    N = wq.shape[0]
    assert wq.shape == (N, K) and bias.numel() == N and sx.numel() == M and sw.numel() == N
    if out is None:
        out = torch.empty(*xq.shape[:-1], N, dtype=bf16, device=xq.device)
    _chk(out, "out")
    assert out.numel() == M * N
    nmod = 0
    if epilogue in (EPI_BIAS_GATE_RES, EPI_BIAS_RES):
        _chk(res, "res")
        assert res.numel() == M * N
    if epilogue == EPI_BIAS_GATE_RES:
        _chk(e, "e")
        nmod = e.shape[-2]
        assert e.shape[-1] == N and e.numel() == (M // frame_len) * nmod * N
        if mod is not None:
            _chk(mod, "mod")
            assert mod.numel() == nmod * N
    lib = _lib.load()
    t0 = _t0(tag)
    _lib.check(lib.ll_gemm_w8a8(xq.data_ptr(), sx.data_ptr(), wq.data_ptr(), sw.data_ptr(), bias.data_ptr(),
                                out.data_ptr(), M, N, K, N, epilogue, _ptr(res), _ptr(e), _ptr(mod), nmod, gate_idx,
                                rows_per_batch, frame_len, _stream()), "ll_gemm_w8a8")
    _t1(tag, t0, 2.0 * M * N * K)
    return out


def linear_small(x, w, bias, act_in: int = 0, act_out: int = 0):
    """Few-row linear (time embedding); rows are processed 8 at a time."""
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape == (N, K) and bias.numel() == N
    out = torch.empty(*x.shape[:-1], N, dtype=bf16, device=x.device)
    lib = _lib.load()
    x2, o2 = x.view(M, K), out.view(M, N)
    for m0 in range(0, M, 8):
        m = min(8, M - m0)
        _lib.check(lib.ll_linear_small(x2[m0:].data_ptr(), w.data_ptr(), bias.data_ptr(), o2[m0:].data_ptr(), m, N, K,
                                       act_in, act_out, _stream()), "ll_linear_small")
    return out


def flash_attn(q, k, v, segments, out=None, scale: Optional[float] = None, tag: Optional[str] = None):
    """q [B,Lq,H,128] (contiguous); k,v [B,Sk,H,128]; keys = concatenation of up to two row ranges
    [(start, end), ...] of k/v.  Returns [B,Lq,H,128]."""
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v")
    B, Lq, H, D = q.shape
    assert D == 128, "kernel is specialised for head_dim 128"
    Sk = k.shape[1]
    assert k.shape == (B, Sk, H, D) and v.shape == k.shape
    segs = [(int(a), int(b)) for a, b in segments if b > a]
    assert 1 <= len(segs) <= 2, segments
    for a, b_ in segs:
        assert 0 <= a < b_ <= Sk, (segments, Sk)
    (s0, e0) = segs[0]
    (s1, e1) = segs[1] if len(segs) == 2 else (0, 0)
    out = torch.empty_like(q) if out is None else _chk(out, "out")
    assert out.shape == q.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    lib = _lib.load()
    t0 = None
    if timer is not None:
        nkeys = (e0 - s0) + (e1 - s1)
        tag = tag or "flash_attn"
        t0 = timer.begin(tag)
    _lib.check(lib.ll_flash_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, Lq, H, H * D, H * D,
                                 H * D, Sk * H * D, s0, e0 - s0, s1, e1 - s1, scale, _stream()),
               "ll_flash_attn")
    if timer is not None:
        timer.end(tag, t0, 4.0 * B * H * Lq * nkeys * D)     # algorithmic FLOPs: QK^T + PV
    return out


def flash_attn_qnorm_ok(H: int, nkeys: int) -> bool:
    return bool(_lib.load().ll_flash_attn_qnorm_ok(H, nkeys))


def flash_attn_qnorm(q, ssq, norm_w, eps: float, k, v, nkeys: int, out=None, scale: Optional[float] = None, tag: Optional[str] = None):
    """Attention over keys [0, nkeys) of k / v with WanRMSNorm(q; norm_w) applied inside the kernel: q [B,Lq,H,128] is the RAW
    projection output and ssq [H, B*Lq] its per-plane row sums of squares (gemm_ssq).  model.py:172,189 in two launches."""
    _chk(q, "q"); _chk(k, "k"); _chk(v, "v"); _chk(norm_w, "norm_w"); _chk(ssq, "ssq", torch.float32)
    B, Lq, H, D = q.shape
    assert D == 128 and norm_w.numel() == H * D and ssq.shape == (H, B * Lq), (q.shape, ssq.shape, norm_w.shape)
    Sk = k.shape[1]
    assert k.shape == (B, Sk, H, D) and v.shape == k.shape and 0 < nkeys <= Sk
    out = torch.empty_like(q) if out is None else _chk(out, "out")
    assert out.shape == q.shape
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    lib = _lib.load()
    tag = tag or "flash_attn"
    t0 = _t0(tag)
    _lib.check(lib.ll_flash_attn_qnorm(q.data_ptr(), ssq.data_ptr(), norm_w.data_ptr(), eps, k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                       B, Lq, H, H * D, H * D, H * D, Sk * H * D, 0, nkeys, scale, _stream()), "ll_flash_attn_qnorm")
    _t1(tag, t0, 4.0 * B * H * Lq * nkeys * D)
    return out


def patchify(x):
    """x [B,F,Cin,H,W] -> [B, F*(H/2)*(W/2), Cin*4]."""
    _chk(x, "x")
    B, F, Cin, H, W = x.shape
    out = torch.empty(B, F * (H // 2) * (W // 2), Cin * 4, dtype=bf16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ll_patchify(x.data_ptr(), out.data_ptr(), B, F, Cin, H, W, _stream()), "ll_patchify")
    return out


def sinusoid(t, dim: int):
    """t float32 [n] -> bf16 [n, dim]."""
    _chk(t, "t", torch.float32)
    n = t.numel()
    out = torch.empty(n, dim, dtype=bf16, device=t.device)
    lib = _lib.load()
    _lib.check(lib.ll_sinusoid(t.data_ptr(), out.data_ptr(), n, dim, _stream()), "ll_sinusoid")
    return out


def unpatchify_x0(head, xt, sigma) -> Tuple[torch.Tensor, torch.Tensor]:
    """head [B, F*h*w, 4*Cout]; xt [B,F,Cout,H,W]; sigma float32 [B*F] -> (flow, x0) [B,F,Cout,H,W]."""
    _chk(head, "head"); _chk(xt, "xt"); _chk(sigma, "sigma", torch.float32)
    B, F, Cout, H, W = xt.shape
    assert head.shape == (B, F * (H // 2) * (W // 2), 4 * Cout), head.shape
    assert sigma.numel() == B * F
    flow, x0 = torch.empty_like(xt), torch.empty_like(xt)
    lib = _lib.load()
    _lib.check(lib.ll_unpatchify_x0(head.data_ptr(), xt.data_ptr(), sigma.data_ptr(), flow.data_ptr(), x0.data_ptr(),
                                    B, F, Cout, H, W, _stream()), "ll_unpatchify_x0")
    return flow, x0


def add_noise(x0, noise, sigma):
    """x0, noise [N, ...]; sigma float32 [N] -> bf16((1-sigma) x0 + sigma noise)."""
    _chk(x0, "x0"); _chk(noise, "noise"); _chk(sigma, "sigma", torch.float32)
    N = x0.shape[0]
    assert noise.shape == x0.shape and sigma.numel() == N
    out = torch.empty_like(x0)
    lib = _lib.load()
    _lib.check(lib.ll_add_noise(x0.data_ptr(), noise.data_ptr(), sigma.data_ptr(), out.data_ptr(), N,
                                x0.numel() // N, _stream()), "ll_add_noise")
    return out


def sigma_lookup(t, timesteps, sigmas):
    """t float32 [n] (device) -> sigmas[argmin |timesteps - t|] float32 [n]."""
    _chk(t, "t", torch.float32); _chk(timesteps, "timesteps", torch.float32); _chk(sigmas, "sigmas", torch.float32)
    assert timesteps.numel() == sigmas.numel()
    out = torch.empty(t.numel(), dtype=torch.float32, device=t.device)
    lib = _lib.load()
    _lib.check(lib.ll_sigma_lookup(t.data_ptr(), timesteps.data_ptr(), sigmas.data_ptr(), out.data_ptr(), t.numel(),
                                   timesteps.numel(), _stream()), "ll_sigma_lookup")
    return out


# ---- VAE decoder (channels-last) -------------------------------------------------------------------------------------
_zero16 = {}


def zero_row(device) -> torch.Tensor:
    """64 zero bytes on `device`: the source of every padding chunk of the implicit-GEMM convolution."""
    key = str(device)
    if key not in _zero16:
        _zero16[key] = torch.zeros(32, dtype=bf16, device=device)
    return _zero16[key]


def pack_conv_weight(w: torch.Tensor, bias: torch.Tensor):
    """Reference conv weight [Cout, Cin, (kt,) kh, kw] -> the kernel's [Cout8, Kpad] bf16 with k = (tap, ci), rows padded
    to a multiple of 8 output channels and K to a multiple of 64 (zeros), plus the padded bias.  One-time, at load."""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    cout, cin, kt, kh, kw = w.shape
    assert kh == kw
    k = kt * kh * kw * cin
    kpad = (k + 63) // 64 * 64
    cout8 = (cout + 7) // 8 * 8
    packed = torch.zeros(cout8, kpad, dtype=bf16, device=w.device)
    packed[:cout, :k] = w.permute(0, 2, 3, 4, 1).reshape(cout, k).to(bf16)
    b = torch.zeros(cout8, dtype=bf16, device=w.device)
    b[:cout] = bias.to(bf16)
    return packed.contiguous(), b, (cin, cout8, kpad, kt, kh)


def conv_cl(x, packed, bias, geo, upsample: bool = False, res=None, out=None):
    """Channels-last convolution.  Temporal kernels (kt == 3): x is [2 + T, H, W, Cin] -- the stream's previous two input
    frames (zeros at its start) followed by the T new ones; otherwise x is [T, H, W, Cin].  packed/bias/geo from
    pack_conv_weight.  Returns [T, H<<up, W<<up, Cout8]."""
    cin, cout, kpad, kt, kh = geo
    _chk(x, "x"); _chk(packed, "w"); _chk(bias, "bias")
    hist = 2 if kt > 1 else 0
    T, H, W, C = x.shape[0] - hist, x.shape[1], x.shape[2], x.shape[3]
    assert T >= 1, "conv_cl: a temporal convolution takes [2 + T, H, W, Cin] (two history frames first)"
    assert C == cin, f"conv_cl: input has {C} channels, weights expect {cin}"
    assert packed.shape == (cout, kpad) and bias.numel() == cout
    Ho, Wo = (2 * H, 2 * W) if upsample else (H, W)
    if out is None:
        out = torch.empty(T, Ho, Wo, cout, dtype=bf16, device=x.device)
    assert out.shape == (T, Ho, Wo, cout) and out.is_contiguous() and out.dtype == bf16
    if res is not None:
        _chk(res, "res")
        assert res.shape == out.shape
    lib = _lib.load()
    t0 = timer.begin("conv") if timer is not None else None
    _lib.check(lib.ll_conv_cl(x[hist:].data_ptr(), zero_row(x.device).data_ptr(), packed.data_ptr(), bias.data_ptr(),
                              _ptr(res), out.data_ptr(), T, H, W, cin, cout, kpad, kt, kh, 1 if upsample else 0, cout,
                              _stream()), "ll_conv_cl")
    if timer is not None:
        timer.end("conv", t0, 2.0 * T * Ho * Wo * cout * (kt * kh * kh * cin))
    return out


def conv_cl_rms_ok(geo, H: int, W: int, upsample: bool = False) -> bool:
    """conv_cl_rms covers this convolution (the halo-tile kernel with every channel of a pixel in one workgroup: Cout = 96)."""
    cin, cout, kpad, kt, kh = geo
    return bool(_lib.load().ll_conv_cl_rms_ok(H, W, cin, cout, kt, kh, 1 if upsample else 0))


def conv_cl_rms(x, packed, bias, geo, gamma, out_rms, silu: bool = True, upsample: bool = False, res=None, want_raw: bool = False):
    """conv_cl + the RMS_norm (+ SiLU) of its result in ONE launch: out_rms [T, Ho, Wo, Cout] = rms_silu(conv(x) [+ res]) -- what
    ResidualBlock feeds its next convolution (vae.py:193-220).  want_raw: also return the un-normalised tensor (the next block's
    shortcut), else None.  Only where conv_cl_rms_ok(...)."""
    cin, cout, kpad, kt, kh = geo
    _chk(x, "x"); _chk(packed, "w"); _chk(bias, "bias"); _chk(gamma, "gamma"); _chk(out_rms, "out_rms")
    hist = 2 if kt > 1 else 0
    T, H, W, C = x.shape[0] - hist, x.shape[1], x.shape[2], x.shape[3]
    assert T >= 1 and C == cin and packed.shape == (cout, kpad) and bias.numel() == cout and gamma.numel() == cout
    Ho, Wo = (2 * H, 2 * W) if upsample else (H, W)
    assert out_rms.shape == (T, Ho, Wo, cout) and out_rms.is_contiguous()
    out = torch.empty(T, Ho, Wo, cout, dtype=bf16, device=x.device) if want_raw else None
    if res is not None:
        _chk(res, "res")
        assert res.shape == out_rms.shape
    lib = _lib.load()
    t0 = timer.begin("conv") if timer is not None else None
    _lib.check(lib.ll_conv_cl_rms(x[hist:].data_ptr(), zero_row(x.device).data_ptr(), packed.data_ptr(), bias.data_ptr(), _ptr(res),
                                  _ptr(out), gamma.data_ptr(), out_rms.data_ptr(), 1 if silu else 0, T, H, W, cin, cout, kpad, kt, kh,
                                  1 if upsample else 0, cout, _stream()), "ll_conv_cl_rms")
    if timer is not None:
        timer.end("conv", t0, 2.0 * T * Ho * Wo * cout * (kt * kh * kh * cin))
    return out


def rms_silu_cl(x, gamma, silu: bool = True, out=None):
    _chk(x, "x"); _chk(gamma, "gamma")
    C = x.shape[-1]
    assert gamma.numel() == C
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.ll_rms_silu_cl(x.data_ptr(), gamma.data_ptr(), out.data_ptr(), x.numel() // C, C, 1 if silu else 0,
                                  _stream()), "ll_rms_silu_cl")
    return out


def softmax_rows(s, scale: float, n_valid: Optional[int] = None, out=None):
    """softmax(scale * s[:, :n_valid]) along the last dim; columns >= n_valid come out as zeros."""
    _chk(s, "s")
    ld = s.shape[-1]
    N = ld if n_valid is None else n_valid
    if out is None:
        out = torch.empty_like(s)
    lib = _lib.load()
    _lib.check(lib.ll_softmax_rows(s.data_ptr(), out.data_ptr(), s.numel() // ld, N, ld, float(scale), _stream()),
               "ll_softmax_rows")
    return out


def vae_unscale_cl(z, mean, inv_std):
    """z [T, C, h, w] bf16 -> bf16(bf16(z / inv_std) + mean) as channels-last [T, h, w, C]."""
    _chk(z, "z"); _chk(mean, "mean"); _chk(inv_std, "inv_std")
    T, C, h, w = z.shape
    assert mean.numel() == C and inv_std.numel() == C
    out = torch.empty(T, h, w, C, dtype=bf16, device=z.device)
    lib = _lib.load()
    _lib.check(lib.ll_vae_unscale_cl(z.data_ptr(), mean.data_ptr(), inv_std.data_ptr(), out.data_ptr(), T, C, h, w, _stream()),
               "ll_vae_unscale_cl")
    return out


def cl_to_tchw_clamp(x):
    """[T,H,W,C>=3] bf16 channels-last -> fp32 [T,3,H,W] clamped to [-1,1]."""
    _chk(x, "x")
    T, H, W, C = x.shape
    out = torch.empty(T, 3, H, W, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.ll_cl_to_tchw_clamp(x.data_ptr(), out.data_ptr(), T, H, W, C, _stream()), "ll_cl_to_tchw_clamp")
    return out


# ---- umT5 text encoder -----------------------------------------------------------------------------------------------
def t5_rmsnorm(x, w, eps: float = 1e-6):
    _chk(x, "x"); _chk(w, "w")
    C = x.shape[-1]
    assert w.numel() == C
    out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.ll_t5_rmsnorm(x.data_ptr(), w.data_ptr(), out.data_ptr(), x.numel() // C, C, float(eps), _stream()),
               "ll_t5_rmsnorm")
    return out


def gemm_res_t5norm(x, w, bias, res, norm_w, eps: float = 1e-6, tag: str = "gemm"):
    """(x_new, h) = (res + (x @ w^T + bias), T5LayerNorm(x_new; norm_w)): umT5's `x = x + linear(.)` and the norm that follows it
    (t5.py:119-160) -- on the small-M path one pass over the rows does the K-range sum, bias, residual and norm."""
    _chk(x, "x"); _chk(w, "w"); _chk(bias, "bias"); _chk(res, "res"); _chk(norm_w, "norm_w")
    K = x.shape[-1]
    M = x.numel() // K
    N = w.shape[0]
    assert w.shape == (N, K) and bias.numel() == N and res.numel() == M * N and norm_w.numel() == N
    out = torch.empty(*x.shape[:-1], N, dtype=bf16, device=x.device)
    h = torch.empty_like(out)
    lib = _lib.load()
    ws = ksplit_workspace(x.device, M, N, K) if _ksplit_plan(lib, M, N, K) else None
    t0 = _t0(tag)
    _lib.check(lib.ll_gemm_bf16_ksplit_t5norm(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, K, N, res.data_ptr(),
                                              norm_w.data_ptr(), float(eps), h.data_ptr(), _ptr(ws), ws.numel() if ws is not None else 0,
                                              _stream()), "ll_gemm_bf16_ksplit_t5norm")
    _t1(tag, t0, 2.0 * M * N * K)
    return out, h


def t5_gated_gelu(h):
    """h [M, 2F] = [gate | fc1] -> bf16(fc1 * GELU_py(gate)) [M, F]."""
    _chk(h, "h")
    M, F2 = h.numel() // h.shape[-1], h.shape[-1]
    assert F2 % 16 == 0
    out = torch.empty(*h.shape[:-1], F2 // 2, dtype=bf16, device=h.device)
    lib = _lib.load()
    _lib.check(lib.ll_t5_gated_gelu(h.data_ptr(), out.data_ptr(), M, F2 // 2, _stream()), "ll_t5_gated_gelu")
    return out


def gather_rows(table, ids):
    """table [V, C] bf16, ids int64 [n] on the device with 0 <= id < V (checked by the caller on the host copy)."""
    _chk(table, "table"); _chk(ids, "ids", torch.int64)
    V, C = table.shape
    out = torch.empty(ids.numel(), C, dtype=bf16, device=table.device)
    lib = _lib.load()
    _lib.check(lib.ll_gather_rows(table.data_ptr(), ids.data_ptr(), out.data_ptr(), ids.numel(), C, V, _stream()),
               "ll_gather_rows")
    return out


def t5_attention(qk, vt, bias_tab, num_heads: int, seq_len: int):
    """qk [L, 2*H*64] = [q | k] rows; vt [H*64, L]; bias_tab [H, 2L-1] -> [L, H*64]."""
    _chk(qk, "qk"); _chk(vt, "vt"); _chk(bias_tab, "bias_tab")
    L, two_c = qk.shape
    C = num_heads * 64
    assert two_c == 2 * C and vt.shape == (C, L) and bias_tab.shape == (num_heads, 2 * L - 1)
    assert 1 <= seq_len <= L
    out = torch.empty(L, C, dtype=bf16, device=qk.device)
    lib = _lib.load()
    _lib.check(lib.ll_t5_attention(qk.data_ptr(), qk[:, C:].data_ptr(), vt.data_ptr(), bias_tab.data_ptr(), out.data_ptr(),
                                   L, num_heads, two_c, C, int(seq_len), _stream()), "ll_t5_attention")
    return out
