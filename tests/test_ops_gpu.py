"""Parity of every HIP kernel (called through the C ABI via longlive_amd.ops) against the CPU oracle on the same
seeded inputs.  Elementwise kernels reproduce the reference's bf16 rounding points, so they are held to <= 1 bf16
ulp with >= 99% of elements bit-exact; MFMA kernels (fp32 accumulation in a different order) to stated tolerances."""
import math

import pytest
import torch

from longlive_amd import synth
from oracle import ref_ops as R
from util import assert_bf16_close, bf, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from longlive_amd import ops as o
    return o


def hn(name, shape, scale=1.0, shift=0.0, seed=101, device="cpu"):
    """bf16 test data from the integer counter hash (bit-identical on the host and on the device: tests whose inputs only ever
    live on the GPU generate them there -- the host takes seconds per 40 M elements)."""
    return (synth.hash_normal(seed, name, shape, device=device) * scale + shift).to(bf)


@pytest.mark.parametrize("C,F,fs,B", [(1536, 3, 40, 1), (256, 2, 24, 2), (1536, 1, 7, 2), (2048, 2, 9, 1), (1024, 1, 5, 1), (520, 2, 6, 1), (8, 1, 3, 1)])
def test_ln_modulate(ops, C, F, fs, B):     # widths: whole 512-column chunks (the template case with every load up front), ragged, one lane
    x = hn("x", (B, F * fs, C), 1.7, 0.3)
    e = hn("e", (B, F, 6, C), 0.5)
    mod = hn("mod", (1, 6, C), 1 / math.sqrt(C))
    ec = (mod.unsqueeze(1) + e).chunk(6, dim=2)                      # causal_model.py:440
    for sh, sc in ((0, 1), (3, 4)):
        want = R.ln_modulate(x, ec[sc], ec[sh], F, 1e-6)
        got = ops.ln_modulate(x.to(DEV), e.to(DEV), mod.view(6, C).to(DEV), sh, sc, F, 1e-6)
        assert_bf16_close(got, want, 1, 0.99, f"ln_modulate {sh},{sc}")


@pytest.mark.parametrize("C,rows", [(1536, 130), (256, 7), (2048, 9), (512, 5), (776, 6), (8, 3)])
def test_layernorm_affine_and_rmsnorm(ops, C, rows):
    x = hn("x2", (rows, C), 2.0, -0.2)
    w = hn("w", (C,), 0.1, 1.0)
    b = hn("b", (C,), 0.1)
    assert_bf16_close(ops.layernorm_affine(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6), R.layer_norm(x, 1e-6, w, b), 1, 0.99,
                      "layernorm_affine")
    assert_bf16_close(ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-6), R.rms_norm(x, w, 1e-6), 1, 0.99, "rmsnorm")
    # strided input (column slice of a wider buffer)
    wide = hn("wide", (rows, 2 * C))
    got = ops.rmsnorm(wide.to(DEV)[:, :C], w.to(DEV), 1e-6)
    assert_bf16_close(got, R.rms_norm(wide[:, :C], w, 1e-6), 1, 0.99, "rmsnorm strided")


@pytest.mark.parametrize("B,F,hp,wp,H,start_frame,ws,ro,wl", [
    (1, 3, 4, 6, 12, 0, 0, 0, 72),        # direct insert at slot 0
    (2, 2, 4, 6, 2, 5, 24, 0, 48),        # later frames, batch 2
    (1, 2, 3, 5, 12, 959, 15, 15, 15),    # recompute with sink protection: first 15 tokens not written
    (1, 1, 30, 52, 12, 7, 100, 0, 1560),  # real frame geometry
    (1, 1, 3, 5, 12, 2, 45, 30, 0),       # recompute entirely inside the protected sink: nothing written (config 1, block 1)
])
def test_qk_norm_rope_kv_store(ops, B, F, hp, wp, H, start_frame, ws, ro, wl):
    D = 128
    C = H * D
    fs = hp * wp
    L = F * fs
    S = ws + wl + 11
    qkv = hn("qkv", (B, L, 3 * C), 1.3)
    wq, wk = hn("wq", (C,), 0.1, 1.0), hn("wk", (C,), 0.1, 1.0)
    freqs = R.make_freqs(D)
    q, k, v = qkv.split(C, dim=-1)
    rq = R.causal_rope_apply(R.rms_norm(q, wq, 1e-6).view(B, L, H, D), (F, hp, wp), freqs, start_frame)
    rk = R.causal_rope_apply(R.rms_norm(k, wk, 1e-6).view(B, L, H, D), (F, hp, wp), freqs, start_frame)
    ck0 = hn("ck", (B, S, H, D))
    cv0 = hn("cv", (B, S, H, D))
    ck, cv = ck0.clone(), cv0.clone()
    ck[:, ws:ws + wl] = rk[:, ro:ro + wl]
    cv[:, ws:ws + wl] = v.reshape(B, L, H, D)[:, ro:ro + wl]

    from longlive_amd.model import CausalWanModelHIP
    m = CausalWanModelHIP(synth.toy_config(num_layers=1), device=DEV)
    rope_f, rope_hw = m._rope_tables(hp, wp, DEV)
    q_out = torch.empty(B, L, C, dtype=bf, device=DEV)
    gk, gv = ck0.to(DEV), cv0.to(DEV)
    ops.qk_norm_rope_kv_store(qkv.to(DEV), wq.to(DEV), wk.to(DEV), rope_f, rope_hw, q_out, gk, gv, D, fs, start_frame,
                              ws, ro, wl, 1e-6)
    assert_bf16_close(q_out.view(B, L, H, D), rq, 1, 0.99, "roped q")
    assert_bf16_close(gk, ck, 1, 0.99, "cache k")
    assert torch.equal(gv.cpu(), cv), "cache v must be a bit-exact copy"
    # slots outside the write window are untouched
    mask = torch.ones(S, dtype=torch.bool); mask[ws:ws + wl] = False
    assert torch.equal(gk.cpu()[:, mask], ck0[:, mask])


@pytest.mark.parametrize("B,S,dst,src,n", [(1, 40, 8, 12, 20), (2, 64, 4, 20, 30), (1, 50, 3, 4, 40), (1, 30, 0, 29, 1)])
def test_kv_roll(ops, B, S, dst, src, n):
    k0, v0 = hn("rk", (B, S, 2, 128)), hn("rv", (B, S, 2, 128))
    k, v = k0.clone(), v0.clone()
    k[:, dst:dst + n] = k0[:, src:src + n]
    v[:, dst:dst + n] = v0[:, src:src + n]
    gk, gv = k0.to(DEV), v0.to(DEV)
    ops.kv_roll(gk, gv, dst, src, n)
    assert torch.equal(gk.cpu(), k) and torch.equal(gv.cpu(), v)


def _lin_ref(x, w, b):
    return x.double() @ w.double().t() + b.double()


@pytest.mark.parametrize("M,N,K", [(200, 256, 128), (4680, 1536, 1536), (130, 4608, 1536), (257, 1536, 8960),
                                   (72, 64, 1536), (1, 128, 64), (512, 1536, 4096)])
def test_gemm_bias(ops, M, N, K):
    x, w, b = hn("gx", (M, K)), hn("gw", (N, K), 1 / math.sqrt(K)), hn("gb", (N,), 0.1)
    got = ops.gemm(x.to(DEV), w.to(DEV), b.to(DEV)).cpu()
    ref = _lin_ref(x, w, b)
    # fp32 accumulation in a different order than any CPU GEMM: hold to 1 bf16 ulp of the fp64 result
    err = (got.double() - ref).abs()
    tol = ref.abs() * 2 ** -7 + 1e-3
    assert (err <= tol).all(), f"max err {err.max()} at |ref| {ref.abs().flatten()[err.argmax()]}"
    assert rel_l2(got, ref) < 3e-3
    # and agree with the oracle's own bf16 linear about as well as bf16 allows
    want = torch.nn.functional.linear(x, w, b)
    assert_bf16_close(got, want, 1, 0.97, "gemm vs F.linear bf16")


def test_gemm_epilogues(ops):
    B, F, fs, C, Nf = 2, 3, 24, 256, 512
    M = B * F * fs
    x = hn("ex", (B, F * fs, C))
    w1, b1 = hn("ew1", (Nf, C), 1 / 16), hn("eb1", (Nf,), 0.1)
    w2, b2 = hn("ew2", (C, Nf), 1 / 22), hn("eb2", (C,), 0.1)
    e = hn("ee", (B, F, 6, C), 0.5)
    mod = hn("emod", (1, 6, C), 0.1)
    res = hn("eres", (B, F * fs, C))
    ec = (mod.unsqueeze(1) + e).chunk(6, dim=2)
    # GELU epilogue (causal_model.py:406-408)
    h = torch.nn.functional.gelu(torch.nn.functional.linear(x, w1, b1), approximate="tanh")
    gh = ops.gemm(x.to(DEV), w1.to(DEV), b1.to(DEV), ops.EPI_BIAS_GELU)
    assert_bf16_close(gh, h, 2, 0.97, "gemm+gelu")   # v_exp/v_rcp sigmoid form of tanh-GELU: <= 2 ulp
    # gate-residual epilogue (causal_model.py:467): x + (y.unflatten(1,(F,fs)) * e[5]).flatten(1,2)
    y = torch.nn.functional.linear(h, w2, b2)
    want = res + (y.unflatten(1, (F, fs)) * ec[5]).flatten(1, 2)
    got = ops.gemm(h.to(DEV), w2.to(DEV), b2.to(DEV), ops.EPI_BIAS_GATE_RES, res=res.to(DEV), e=e.to(DEV),
                   mod=mod.view(6, C).to(DEV), gate_idx=5, rows_per_batch=F * fs, frame_len=fs)
    assert_bf16_close(got, want, 1, 0.97, "gemm+gate-residual")
    # in-place residual (out aliases res) and plain residual epilogue (causal_model.py:460)
    r2 = res.to(DEV).clone()
    ops.gemm(h.to(DEV), w2.to(DEV), b2.to(DEV), ops.EPI_BIAS_RES, out=r2, res=r2)
    assert_bf16_close(r2, res + y, 1, 0.97, "gemm+residual in place")


def test_linear_small_time_embedding(ops):
    C, fd = 256, 256
    t = torch.tensor([1000.0, 937.5, 833.3333, 625.0, 0.0, 3.0, 17.0, 999.0, 500.0, 250.0, 42.0, 7.0])   # 12 rows > 8
    emb = R.sinusoidal_embedding_1d(fd, t).to(bf)
    got_emb = ops.sinusoid(t.to(DEV), fd)
    assert_bf16_close(got_emb, emb, 1, 0.98, "sinusoid")
    w0, b0 = hn("tw0", (C, fd), 0.05), hn("tb0", (C,), 0.02)
    w2, b2 = hn("tw2", (C, C), 0.05), hn("tb2", (C,), 0.02)
    w3, b3 = hn("tw3", (6 * C, C), 0.05), hn("tb3", (6 * C,), 0.02)
    lin, silu = torch.nn.functional.linear, torch.nn.functional.silu
    e_ref = lin(silu(lin(emb, w0, b0)), w2, b2)
    e0_ref = lin(silu(e_ref), w3, b3)
    d = lambda a: a.to(DEV)
    h = ops.linear_small(d(emb), d(w0), d(b0), act_out=1)
    e = ops.linear_small(h, d(w2), d(b2))
    e0 = ops.linear_small(e, d(w3), d(b3), act_in=1)
    assert_bf16_close(e, e_ref, 2, 0.90, "time e")
    assert_bf16_close(e0, e0_ref, 2, 0.85, "time e0")


@pytest.mark.parametrize("B,Lq,H,Sk,segs", [
    (1, 40, 2, 72, [(0, 72)]),
    (2, 200, 2, 300, [(0, 64), (100, 300)]),          # sink + non-adjacent window, both with ragged tails
    (1, 130, 12, 512, [(0, 512)]),                      # cross-attention shape
    (1, 1560, 12, 4680, [(0, 1560), (1560, 4680)]),     # adjacent ranges merge; real head count
    (1, 33, 1, 7, [(0, 7)]),                            # fewer keys than one tile
    (2, 300, 3, 1500, [(0, 1437)]),                     # >= 1024 keys: ping-pong kernel; batch 2, ragged keys, padded waves
    (1, 64, 1, 1024, [(0, 1024)]),                      # exactly at the ping-pong threshold, all but two waves padding
    (1, 257, 2, 2000, [(37, 1100)]),                    # ping-pong with a key range that does not start at slot 0
])
def test_flash_attn(ops, B, Lq, H, Sk, segs):
    q = hn("aq", (B, Lq, H, 128))
    k = hn("ak", (B, Sk, H, 128))
    v = hn("av", (B, Sk, H, 128), 0.7)
    kk = torch.cat([k[:, a:b] for a, b in segs], 1)
    vv = torch.cat([v[:, a:b] for a, b in segs], 1)
    exact = R.attention_exact(q, kk, vv)
    oracle = R.attention(q, kk, vv).double()
    got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), segs).cpu().double()
    err, oerr = (got - exact).abs().max().item(), (oracle - exact).abs().max().item()
    assert err < 1.2e-2, f"max abs err vs fp64 {err} (oracle's own bf16 path: {oerr})"
    assert rel_l2(got, exact) < 6e-3, (rel_l2(got, exact), rel_l2(oracle, exact))


def _set_tuning(key, value):
    from longlive_amd import _lib
    _lib.check(_lib.load().ll_set_tuning(key.encode(), int(value)), "ll_set_tuning")


def test_flash_attn_online_softmax_rescale(ops):
    """Forces the running-max update late in the key sequence (a spiked key in the last tile)."""
    B, Lq, H, Sk = 1, 64, 1, 256
    q, k, v = hn("sq", (B, Lq, H, 128)), hn("sk", (B, Sk, H, 128)), hn("sv", (B, Sk, H, 128))
    k[0, 250, 0] = (q[0, 5, 0].float() * 3).to(bf)      # huge score for query 5 at key 250
    k[0, 3, 0] = (q[0, 9, 0].float() * 3).to(bf)        # and an early spike for query 9
    exact = R.attention_exact(q, k, v)
    got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [(0, Sk)]).cpu().double()
    assert (got - exact).abs().max().item() < 2e-2


@pytest.mark.parametrize("form", [1])
@pytest.mark.parametrize("B,Lq,H,Sk,seg", [
    (2, 300, 3, 1500, (0, 1437)),         # batch 2, ragged last key tile (29 keys), padded rows + idle waves in the last q-tile
    (1, 64, 1, 1024, (0, 1024)),          # exactly at the threshold: one wave with rows, three idle
    (1, 257, 2, 2000, (37, 1100)),        # key range that does not start at slot 0 and ENDS before the tensor does
    (1, 520, 12, 1300, (0, 1300)),        # real head count, 3 q-tiles; the key range ends exactly at the end of the tensor
    (1, 72, 2, 1081, (0, 1081)),          # last tile holds 57 keys; 72 rows = the last q-tile of Lq 4680
    (2, 300, 12, 512, (0, 512)),          # cross-attention's range: exactly attn_asm_min_keys (8 tiles), batch 2
    (1, 130, 2, 700, (64, 633)),          # 569 keys: 9 tiles, the last one with 57 keys; shorter than the loop's unroll + prologue
])
def test_flash_attn_asm_kernel(ops, form, B, Lq, H, Sk, seg):
    """flash_attn_asm_kernel (tuning key attn_asm): the generated one-wave-per-SIMD kernel against fp64 and against the shipped kernel."""
    q = hn("aq", (B, Lq, H, 128))
    k = hn("ak", (B, Sk, H, 128))
    v = hn("av", (B, Sk, H, 128), 0.7)
    exact = R.attention_exact(q, k[:, seg[0]:seg[1]], v[:, seg[0]:seg[1]])
    try:
        _set_tuning("attn_asm", 0)
        base = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [seg]).cpu()
        _set_tuning("attn_asm", form)
        got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [seg]).cpu()
    finally:
        _set_tuning("attn_asm", 1)
    assert torch.isfinite(got.float()).all()
    err, berr = (got.double() - exact).abs().max().item(), (base.double() - exact).abs().max().item()
    assert err < 1.2e-2 and rel_l2(got, exact) < 6e-3, (err, berr, rel_l2(got, exact))
    assert (got.float() - base.float()).abs().max().item() < 8e-3


@pytest.mark.parametrize("form", [1])
def test_flash_attn_asm_kernel_rescale(ops, form):
    """Running-max jumps far above the lazy-max threshold (late, early, mid-range; both q-blocks of a wave, several waves): the
    rescale path of the generated kernel (O, l, the -m tile and the pending score tile, once, after the pending P.V)."""
    B, Lq, H, Sk = 1, 256, 1, 2048
    q, k, v = hn("sq", (B, Lq, H, 128)), hn("sk", (B, Sk, H, 128)), hn("sv", (B, Sk, H, 128))
    for qi, ki in ((5, 2040), (9, 3), (37, 1000), (100, 77), (200, 1999), (255, 1024)):
        k[0, ki, 0] = (q[0, qi, 0].float() * 3).to(bf)
    exact = R.attention_exact(q, k, v)
    try:
        _set_tuning("attn_asm", form)
        got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [(0, Sk)]).cpu().double()
    finally:
        _set_tuning("attn_asm", 1)
    assert (got - exact).abs().max().item() < 2e-2


def test_patchify_unpatchify_x0_add_noise(ops):
    cfg = synth.toy_config()
    B, F, Cc, H, W = 2, 3, 16, 8, 12
    x = hn("px", (B, F, Cc, H, W))
    w = hn("pw", (64, Cc, 1, 2, 2), 0.2)
    bias = hn("pb", (64,), 0.1)
    want = torch.nn.functional.conv3d(x.permute(0, 2, 1, 3, 4), w, bias, stride=(1, 2, 2)).flatten(2).transpose(1, 2)
    got = ops.gemm(ops.patchify(x.to(DEV)), w.view(64, -1).to(DEV), bias.to(DEV))
    assert_bf16_close(got, want, 1, 0.97, "patch embedding")
    # unpatchify + flow -> x0
    sch = R.FlowMatchSchedulerRef(5.0)
    head = hn("hd", (B, F * 4 * 6, 64))
    t = torch.tensor([[1000.0, 937.5, 625.0], [0.0, 833.3333, 500.0]])
    flow_ref = torch.einsum("bfhwpqrc->bcfphqwr", head.view(B, F, 4, 6, 1, 2, 2, 16)).reshape(B, 16, F, 8, 12)
    flow_ref = flow_ref.permute(0, 2, 1, 3, 4)
    x0_ref = R.flow_to_x0(sch, flow_ref.flatten(0, 1), x.flatten(0, 1), t.flatten()).unflatten(0, (B, F))
    from longlive_amd.scheduler import FlowMatchScheduler
    ps = FlowMatchScheduler(5.0)
    assert torch.equal(ps.sigmas, sch.sigmas) and torch.equal(ps.timesteps, sch.timesteps)
    sigma = ps.sigma_of(t.to(DEV))
    tid = torch.argmin((sch.timesteps.double().unsqueeze(0) - t.flatten().double().unsqueeze(1)).abs(), dim=1)
    assert torch.equal(sigma.cpu(), sch.sigmas[tid])
    flow, x0 = ops.unpatchify_x0(head.to(DEV), x.to(DEV), sigma)
    assert torch.equal(flow.cpu(), flow_ref)
    nd = (x0.cpu() != x0_ref).sum().item()
    assert nd == 0, f"fp64 flow->x0 must be bit-exact: {nd} of {x0_ref.numel()} differ, e.g. {x0.cpu()[x0.cpu() != x0_ref][:4]} vs {x0_ref[x0.cpu() != x0_ref][:4]}"
    nz = hn("nz", (B * F, Cc, H, W))
    tt = torch.tensor([937.5, 833.3333, 625.0, 625.0, 0.0, 1000.0])
    want = sch.add_noise(x0_ref.flatten(0, 1), nz, tt)
    got = ps.add_noise(x0.flatten(0, 1), nz.to(DEV), tt.to(DEV))
    assert torch.equal(got.cpu(), want), "add_noise must be bit-exact"


def _dequant_check(q, sc, ref_bf16, what):
    """q * scale must reproduce the bf16 tensor to within half a quantisation step per row."""
    deq = q.cpu().float() * sc.cpu().view(-1, 1)
    ref = ref_bf16.float().reshape(deq.shape)
    step = ref.abs().amax(dim=1, keepdim=True) / 127.0
    assert ((deq - ref).abs() <= 0.5 * step + 1e-6).all(), what
    assert torch.allclose(sc.cpu(), (ref.abs().amax(dim=1) / 127.0).clamp_min(0) + (ref.abs().amax(dim=1) == 0).float(), rtol=1e-6), what


def test_int8_quantize_and_w8a8_gemm(ops):
    M, N, K = 333, 264, 384
    x, w, b = hn("qx", (M, K), 1.5), hn("qw", (N, K), 0.05), hn("qb", (N,), 0.1)
    x[17] = 0                                                     # an all-zero row keeps scale 1
    xq, sx = ops.quantize_rows(x.to(DEV))
    wq, sw = ops.quantize_rows(w.to(DEV))
    _dequant_check(xq, sx, x, "activation quantisation")
    _dequant_check(wq, sw, w, "weight quantisation")
    # the CPU oracle's restatement of the scheme (oracle/ref_model.py RefModel.quantize_rows / lin with quant="int8"): identical
    # codes and scales on identical inputs, bit for bit
    from oracle.ref_model import RefModel
    for t, tq, ts in ((x, xq, sx), (w, wq, sw)):
        oq, osc = RefModel.quantize_rows(t)
        assert torch.equal(tq.cpu().double(), oq) and torch.equal(ts.cpu(), osc)
    got = ops.gemm_w8a8(xq, sx, wq, sw, b.to(DEV)).cpu()
    # exact integer reference of the same quantised operands
    acc = xq.cpu().to(torch.int64) @ wq.cpu().to(torch.int64).t()
    want = (acc.double() * (sx.cpu().double().view(-1, 1) * sw.cpu().double().view(1, -1)) + b.double()).float().to(bf)
    assert_bf16_close(got, want, 1, 0.97, "w8a8 gemm vs exact integer reference")
    # and close to the unquantised product
    ref = x.double() @ w.double().t() + b.double()
    assert rel_l2(got, ref) < 2e-2
    # fused epilogues share code with the bf16 kernel: check the residual one
    res = hn("qres", (M, N))
    got2 = ops.gemm_w8a8(xq, sx, wq, sw, b.to(DEV), ops.EPI_BIAS_RES, res=res.to(DEV))
    assert_bf16_close(got2, res + want, 1, 0.97, "w8a8 gemm + residual")


def test_fused_ln_quantisation_is_bit_identical_to_unfused(ops):
    C, F, fs, B = 1536, 3, 40, 1
    x = hn("x", (B, F * fs, C), 1.7, 0.3, device=DEV)
    e = hn("e", (B, F, 6, C), 0.5, device=DEV)
    mod = hn("mod", (6, C), 1 / math.sqrt(C), device=DEV)
    q1, s1 = ops.ln_modulate_q8(x, e, mod, 0, 1, F, 1e-6)
    q2, s2 = ops.quantize_rows(ops.ln_modulate(x, e, mod, 0, 1, F, 1e-6))
    assert torch.equal(q1, q2) and torch.equal(s1, s2)
    w, b = hn("w", (C,), 0.1, 1.0, device=DEV), hn("b", (C,), 0.1, device=DEV)
    q1, s1 = ops.layernorm_affine_q8(x, w, b, 1e-6)
    q2, s2 = ops.quantize_rows(ops.layernorm_affine(x, w, b, 1e-6))
    assert torch.equal(q1, q2) and torch.equal(s1, s2)


def test_modulation_table_paths_are_bit_identical(ops):
    """`modulation + e` evaluated once per (layer, frame) by ll_modulation_table and handed to ln_modulate / the gate-residual
    GEMM epilogue with mod=None gives the bits of the per-row evaluation (same bf16 rounding points)."""
    B, F, fs, C, NL = 2, 3, 24, 256, 3
    x = hn("mx", (B, F * fs, C), 1.7, 0.3, device=DEV)
    e = hn("me", (B, F, 6, C), 0.5, device=DEV)
    mods = hn("mmods", (NL, 6, C), 0.1, device=DEV)
    tab = ops.modulation_table(e, mods)
    assert tab.shape == (NL, B, F, 6, C)
    assert torch.equal(tab.cpu(), (mods.cpu().view(NL, 1, 1, 6, C) + e.cpu().unsqueeze(0)).to(bf))
    for l in range(NL):
        for sh, sc in ((0, 1), (3, 4)):
            a = ops.ln_modulate(x, e, mods[l], sh, sc, F, 1e-6)
            b = ops.ln_modulate(x, tab[l], None, sh, sc, F, 1e-6)
            assert torch.equal(a, b)
        qa, sa = ops.ln_modulate_q8(x, e, mods[l], 0, 1, F, 1e-6)
        qb, sb = ops.ln_modulate_q8(x, tab[l], None, 0, 1, F, 1e-6)
        assert torch.equal(qa, qb) and torch.equal(sa, sb)
    h = hn("mh", (B, F * fs, 512), device=DEV)
    w, bias, res = hn("mw", (C, 512), 1 / 22, device=DEV), hn("mb", (C,), 0.1, device=DEV), hn("mres", (B, F * fs, C), device=DEV)
    a = ops.gemm(h, w, bias, ops.EPI_BIAS_GATE_RES, res=res, e=e, mod=mods[1], gate_idx=5, rows_per_batch=F * fs, frame_len=fs)
    b = ops.gemm(h, w, bias, ops.EPI_BIAS_GATE_RES, res=res, e=tab[1], mod=None, gate_idx=5, rows_per_batch=F * fs, frame_len=fs)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,F,fs,C,NL", [(2, 3, 24, 256, 3), (1, 3, 1560, 1536, 2), (1, 2, 35, 1280, 1), (1, 1, 5, 2048, 1), (1, 2, 3, 8, 2)])
def test_modulation_table_f32_and_ln_modulate_tab_are_bit_identical(ops, B, F, fs, C, NL):
    """The fp32 table (chunks 1 and 4 hold 1 + scale, rounded where the reference rounds) + ll_ln_modulate_tab give the bits of
    ll_ln_modulate / ll_ln_modulate_q8 on the bf16 table -- production width (all chunks in the row), a ragged width, a small one."""
    x = hn("tx", (B, F * fs, C), 1.7, 0.3, device=DEV)
    e = hn("te", (B, F, 6, C), 0.5, device=DEV)
    mods = hn("tmods", (NL, 6, C), 0.1, device=DEV)
    tab = ops.modulation_table(e, mods)
    t32 = ops.modulation_table_f32(e, mods, 0b010010)
    assert t32.shape == (NL, B, F, 6, C) and t32.dtype == torch.float32
    want = tab.float()
    want[:, :, :, [1, 4]] = (1.0 + want[:, :, :, [1, 4]]).to(bf).float()
    assert torch.equal(t32, want)
    for l in range(NL):
        for sh, sc in ((0, 1), (3, 4)):
            assert torch.equal(ops.ln_modulate_tab(x, t32[l], sh, sc, F, 1e-6), ops.ln_modulate(x, tab[l], None, sh, sc, F, 1e-6))
            qa, sa = ops.ln_modulate_tab(x, t32[l], sh, sc, F, 1e-6, q8=True)
            qb, sb = ops.ln_modulate_q8(x, tab[l], None, sh, sc, F, 1e-6)
            assert torch.equal(qa, qb) and torch.equal(sa, sb)


@pytest.mark.parametrize("B,F,hp,wp,H,K,ws,ro,wl", [
    (1, 3, 4, 6, 2, 256, 7, 0, 72),        # all new tokens inserted
    (2, 2, 4, 6, 2, 256, 24, 9, 30),       # batch 2, sink-protected head (roped_offset) and a short window
    (1, 1, 3, 5, 12, 1536, 45, 30, 0),     # nothing inserted (recompute inside the sink)
    (1, 3, 30, 52, 12, 1536, 14040, 0, 4680),   # the steady-state launch: 4680 x 4608 x 1536 (256x192 tiles), last 3 frames of the window
    (1, 2, 10, 13, 3, 512, 50, 70, 101),   # generated 192-wide kernel (V third at column 768): tokens 70..170 of 260 inserted, ragged tile rows
    (1, 3, 30, 52, 12, 1536, 3000, 1560, 3120),   # shipped shape, first frame protected (roped_offset = one frame)
])
def test_qkv_projection_with_v_insert_is_bit_identical(ops, B, F, hp, wp, H, K, ws, ro, wl):
    """ll_gemm_bf16_qkv (V third written into the cache by the GEMM epilogue) + ll_qk_norm_rope_kv_store(cache_v=NULL) against
    the unfused pair: q, the K cache and the V cache must not differ by a bit, untouched slots stay untouched.  Also W8A8."""
    D = 128
    C = H * D
    fs, L = hp * wp, F * hp * wp
    S = ws + wl + 11
    x = hn("vx", (B, L, K), device=DEV)
    w, b = hn("vw", (3 * C, K), 1 / math.sqrt(K), device=DEV), hn("vb", (3 * C,), 0.1, device=DEV)
    wq, wk = hn("wq", (C,), 0.1, 1.0, device=DEV), hn("wk", (C,), 0.1, 1.0, device=DEV)
    from longlive_amd.model import CausalWanModelHIP
    m = CausalWanModelHIP(synth.toy_config(num_layers=1), device=DEV)
    rope_f, rope_hw = m._rope_tables(hp, wp, DEV)
    ck0, cv0 = hn("ck", (B, S, H, D), device=DEV), hn("cv", (B, S, H, D), device=DEV)

    def run(fused, int8):
        ck, cv = ck0.clone(), cv0.clone()
        q = torch.empty(B, L, C, dtype=bf, device=DEV)
        if int8:
            xq, sx = ops.quantize_rows(x)
            wq8, sw = ops.quantize_rows(w)
        if fused:
            qkv = (ops.gemm_qkv_v_insert(None, (wq8, sw), b, cv, ws, ro, wl, xq=(xq, sx)) if int8
                   else ops.gemm_qkv_v_insert(x, w, b, cv, ws, ro, wl))
            ops.qk_norm_rope_kv_store(qkv, wq, wk, rope_f, rope_hw, q, ck, None, D, fs, 3, ws, ro, wl, 1e-6)
        else:
            qkv = ops.gemm_w8a8(xq, sx, wq8, sw, b) if int8 else ops.gemm(x, w, b)
            ops.qk_norm_rope_kv_store(qkv, wq, wk, rope_f, rope_hw, q, ck, cv, D, fs, 3, ws, ro, wl, 1e-6)
        return q, ck, cv

    # bit-identity holds between kernels of ONE family (same order of the fp32 sum): the shipped shape runs the generated 192-wide
    # kernel fused and unfused; where the toy shapes would pair a generated kernel with a HIP one, both run the HIP kernels
    from longlive_amd import _lib
    import ctypes as C_
    buf = C_.create_string_buffer(256)
    fam = []
    for plain in (2 if B == 1 else 0, 1):
        _lib.check(_lib.load().ll_gemm_plan_epi(B * L, 3 * C, K, 0, ops.EPI_BIAS, plain, buf, 256), "plan")
        fam.append("asm" if (b"gemm_asm_" in buf.value or b"gemm_asmp_" in buf.value) else "hip")
    if B * L * 3 * C <= (1 << 22) and _lib.load().ll_gemm_ksplit_plan(B * L, 3 * C, K) >= 2:
        fam[1] = "asm, K-split"                         # the unfused projection of few rows sums K in ranges: another order
    if (B, F, hp) == (1, 3, 30):
        assert fam == ["asm", "asm"], "the steady-state QKV launch takes the generated kernel"
    try:
        if fam[0] != fam[1]:
            _set_tuning("gemm_asm", 0)
        for int8 in (False, True):
            if int8 and K % 128:
                continue
            a, bq = run(False, int8), run(True, int8)
            for u, v_, nm in zip(a, bq, ("q", "cache k", "cache v")):
                assert torch.equal(u, v_), f"{nm} differs (int8={int8})"
            mask = torch.ones(S, dtype=torch.bool); mask[ws:ws + wl] = False
            assert torch.equal(bq[2][:, mask], cv0[:, mask])
    finally:
        _set_tuning("gemm_asm", 35)


@pytest.mark.parametrize("M,N,K,epi", [(4680, 8960, 1536, "gelu"), (4680, 1536, 1536, "gate"), (4680, 1536, 1536, "res"),
                                       (4680, 1536, 1536, "bias"), (4680, 1536, 8960, "gate"), (300, 224, 256, "gelu"), (512, 20480, 4096, "bias"),
                                       (70, 128, 256, "bias"), (9360, 1536, 1536, "gate"), (513, 448, 320, "gelu")])
def test_gemm_asm_kernels_match_hip_kernels(ops, M, N, K, epi):
    """The generated one-wave-per-SIMD GEMM kernels (tuning key gemm_asm; gen/gemm_asm_gen.py) against the HIP kernels they replace,
    at the block linears' shapes (FFN1, O / cross-o / cross-q, FFN2, B = 2) and at ragged edges: same products, the fp32 sum taken in
    another order -> <= 1-2 bf16 ulp apart (absolute bound where a residual cancels), and against fp64."""
    x = hn("gx", (M, K), device=DEV)
    w = (hn("gw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("gb", (N,), 0.1, device=DEV)
    kw = {}
    code = {"bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS_GELU, "gate": ops.EPI_BIAS_GATE_RES, "res": ops.EPI_BIAS_RES}[epi]
    if epi in ("gate", "res"):
        kw["res"] = hn("gr", (M, N), device=DEV)
    if epi == "gate":
        F_ = 3 if M % 3 == 0 else 1
        kw.update(e=hn("ge", (1, F_, 6, N), 0.5, device=DEV), mod=None, gate_idx=5, rows_per_batch=M, frame_len=M // F_)
    from longlive_amd import _lib
    import ctypes as C
    buf = C.create_string_buffer(256)
    _lib.check(_lib.load().ll_gemm_plan_epi(M, N, K, 0, code, 1, buf, 256), "plan")
    assert b"gemm_asm_" in buf.value or b"gemm_asmp_" in buf.value, buf.value           # the shipped tuning takes the generated kernel for this call
    try:
        _set_tuning("gemm_asm", 0)
        _lib.check(_lib.load().ll_gemm_plan_epi(M, N, K, 0, code, 1, buf, 256), "plan")
        assert b"gemm_kernel_v" in buf.value, buf.value
        want = ops.gemm(x, w, b, code, **kw)
    finally:
        _set_tuning("gemm_asm", 35)
    got = ops.gemm(x, w, b, code, **kw)
    again = ops.gemm(x, w, b, code, **kw)
    assert torch.equal(got, again)
    fused = epi in ("gate", "res")
    assert_bf16_close(got, want, 2, 0.97, f"gemm_asm {M}x{N}x{K} {epi}", atol=4e-2 if fused else None)
    if M * N * K < 2e9:
        ref = (x.double() @ w.double().t() + b.double()).cpu()
        if epi == "bias":
            assert rel_l2(got.cpu(), ref) < 4e-3


@pytest.mark.parametrize("M,N,K,epi", [(4680, 8960, 1536, "gelu"),      # FFN1: 760 tiles on 256 CUs, 2.97 rounds
                                       (18720, 1536, 1536, "gate"),     # the recache forward's O projection: 888 tiles
                                       (18720, 1536, 8960, "gate"),     # ... FFN2
                                       (18720, 1536, 1536, "res"), (18720, 1536, 1536, "bias"),
                                       (9360, 8960, 1536, "gelu"),      # B = 2
                                       (2100, 3584, 256, "gelu")])      # ragged last m-tile (52 rows), shortest K (4 K-steps), 9 x 16 = 144 tiles: classic
def test_gemm_asm_persistent_form_is_the_classic_form_bit_for_bit(ops, M, N, K, epi):
    """Launches with more tiles than CUs run the PERSISTENT generated kernels (gemm_asmp_*: one workgroup per CU walks its tiles and
    stages the next tile's first pieces under the current epilogue; tuning key gemm_asm bit 5).  Same tiles, same K order, same
    epilogue text: the output must equal the classic one-tile-per-workgroup kernels' bit for bit."""
    x = hn("px2", (M, K), device=DEV)
    w = (hn("pw2", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("pb2", (N,), 0.1, device=DEV)
    kw = {}
    code = {"bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS_GELU, "gate": ops.EPI_BIAS_GATE_RES, "res": ops.EPI_BIAS_RES}[epi]
    if epi in ("gate", "res"):
        kw["res"] = hn("pr2", (M, N), device=DEV)
    if epi == "gate":
        kw.update(e=hn("pe2", (1, 3, 6, N), 0.5, device=DEV), mod=None, gate_idx=2, rows_per_batch=M, frame_len=M // 3)
    from longlive_amd import _lib
    import ctypes as C
    buf = C.create_string_buffer(320)
    _lib.check(_lib.load().ll_gemm_plan_epi(M, N, K, 0, code, 1, buf, 320), "plan")
    tiles = ((M + 255) // 256) * (N // (224 if epi == "gelu" else 128))
    assert (b"gemm_asmp_" in buf.value) == (tiles > 256), buf.value
    got = ops.gemm(x, w, b, code, **kw)
    again = ops.gemm(x, w, b, code, **kw)
    try:
        _set_tuning("gemm_asm", 3)
        _lib.check(_lib.load().ll_gemm_plan_epi(M, N, K, 0, code, 1, buf, 320), "plan")
        assert b"gemm_asm_" in buf.value, buf.value
        want = ops.gemm(x, w, b, code, **kw)
    finally:
        _set_tuning("gemm_asm", 35)
    assert torch.equal(got, again) and torch.equal(got, want), (got.float() - want.float()).abs().max().item()


@pytest.mark.parametrize("B,L,ws,ro,wl", [(1, 4680, 14040, 0, 4680), (1, 4680, 4680, 3000, 1680), (1, 18720, 0, 0, 18720), (1, 4680, 4680, 4680, 0)])
def test_qkv_persistent_form_is_the_classic_form_bit_for_bit(ops, B, L, ws, ro, wl):
    """The fused QKV projection (V third redirected into the KV cache) on the persistent 192-wide kernel against the classic one: the
    q | k thirds and the whole cache, bit for bit -- steady state, a partial insert window, the recache forward (L = 18720) and a
    pass that inserts nothing (every V tile skipped)."""
    C, K, S = 1536, 1536, 18720
    x = hn("qx2", (B, L, K), device=DEV)
    w = (hn("qw2", (3 * C, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("qb2", (3 * C,), 0.1, device=DEV)
    cv0 = hn("qc2", (B, S, 12, 128), device=DEV)

    def run():
        cv = cv0.clone()
        out = ops.gemm_qkv_v_insert(x, w, b, cv, ws, ro, wl)
        return out[..., :2 * C].clone(), cv

    got = run()
    try:
        _set_tuning("gemm_asm", 3)
        want = run()
    finally:
        _set_tuning("gemm_asm", 35)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    if wl == 0:
        assert torch.equal(got[1], cv0)


@pytest.mark.parametrize("M,N,K,epi", [(512, 4096, 4096, "res"), (512, 8192, 4096, "bias"), (4096, 512, 4096, "bias"),
                                       (512, 4096, 10240, "res"), (77, 1536, 4096, "bias"), (300, 256, 1024, "res")])
def test_gemm_small_m_split_k_matches_plain(ops, M, N, K, epi):
    """ll_gemm_bf16_ksplit (umT5's linears at 512 tokens, the text K / V projections): K cut into ranges on the generated 256 x 128
    kernel, fp32 tile sums added in a fixed order -> equal to the unsplit HIP kernel up to the order of the fp32 sum, bit-identical
    run to run; ops.gemm takes it by itself for these shapes."""
    from longlive_amd import _lib
    lib = _lib.load()
    S = lib.ll_gemm_ksplit_plan(M, N, K)
    assert S >= 2, (M, N, K, S)
    x = hn("kx", (M, K), device=DEV)
    w = (hn("kw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("kb", (N,), 0.1, device=DEV)
    kw = {"res": hn("kr", (M, N), device=DEV)} if epi == "res" else {}
    code = ops.EPI_BIAS_RES if epi == "res" else ops.EPI_BIAS
    got = ops.gemm(x, w, b, code, **kw)
    again = ops.gemm(x, w, b, code, **kw)
    assert torch.equal(got, again)
    try:
        _set_tuning("gemm_asm", 0)                       # plan -> 0: the HIP kernels, unsplit
        assert lib.ll_gemm_ksplit_plan(M, N, K) == 0
        want = ops.gemm(x, w, b, code, **kw)
    finally:
        _set_tuning("gemm_asm", 35)
    assert_bf16_close(got, want, 2, 0.97, f"small-M split-K {M}x{N}x{K} {epi}", atol=4e-2 if epi == "res" else None)
    ref = (x.double() @ w.double().t() + b.double()).cpu()
    if epi == "bias":
        assert rel_l2(got.cpu(), ref) < 4e-3


def test_gemm_small_m_split_k_is_batch_invariant(ops):
    """The number of K-ranges depends on N and K only: the text K / V projection of two prompts batched (M = 1024) gives each
    prompt the bits it gets alone (M = 512) -- what keeps the B = 2 throughput mode bit-identical to two streams."""
    from longlive_amd import _lib
    lib = _lib.load()
    N, K = 1536, 1536
    assert lib.ll_gemm_ksplit_plan(512, N, K) == lib.ll_gemm_ksplit_plan(1024, N, K) >= 2
    x = hn("bx", (1024, K), device=DEV)
    w = (hn("bw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("bb", (N,), 0.1, device=DEV)
    both = ops.gemm(x, w, b)
    for i in (0, 1):
        assert torch.equal(both[512 * i:512 * (i + 1)], ops.gemm(x[512 * i:512 * (i + 1)].contiguous(), w, b))


@pytest.mark.parametrize("M,N,K", [(512, 4096, 4096), (512, 4096, 10240), (77, 1024, 2048), (300, 1536, 1024)])
def test_gemm_residual_t5norm_is_the_two_kernels(ops, M, N, K):
    """ll_gemm_bf16_ksplit_t5norm: x_new = res + linear(x), h = T5LayerNorm(x_new) in one pass over the rows on the small-M path --
    both outputs bit-identical to ops.gemm(EPI_BIAS_RES) + ops.t5_rmsnorm (the fused pass keeps their order of operations)."""
    x = hn("tx", (M, K), device=DEV)
    w = (hn("tw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("tb", (N,), 0.1, device=DEV)
    res = hn("tr", (M, N), device=DEV)
    nw = hn("tn", (N,), 0.2, 1.0, device=DEV)
    want_x = ops.gemm(x, w, b, ops.EPI_BIAS_RES, res=res)
    want_h = ops.t5_rmsnorm(want_x, nw)
    got_x, got_h = ops.gemm_res_t5norm(x, w, b, res, nw)
    assert torch.equal(got_x, want_x), (got_x.float() - want_x.float()).abs().max().item()
    assert torch.equal(got_h, want_h), (got_h.float() - want_h.float()).abs().max().item()


@pytest.mark.parametrize("M,N,K,epi", [(4680, 8960, 1536, "gelu"), (4680, 1536, 8960, "gate"), (4680, 1536, 1536, "gate"),
                                       (4680, 1536, 1536, "res"), (4680, 1536, 1536, "bias"), (4680, 4608, 1536, "bias"),
                                       (300, 224, 512, "gelu"), (70, 128, 640, "bias"), (9360, 1536, 1536, "gate")])
def test_gemm_asm_w8a8_equals_the_hip_w8a8_kernels(ops, M, N, K, epi):
    """The W8A8 variants of the generated kernels (tuning gemm_asm bit 4; v_mfma_i32_32x32x32_i8, scales in the epilogue): integer
    sums are exact and the epilogue applies gemm_common.h's operations in its order; only the fp32 exp / rcp of the GELU may round
    differently from the compiler's code."""
    x = hn("ix", (M, K), device=DEV)
    w = (hn("iw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("ib", (N,), 0.1, device=DEV)
    xq, sx = ops.quantize_rows(x)
    wq, sw = ops.quantize_rows(w)
    kw = {}
    code = {"bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS_GELU, "gate": ops.EPI_BIAS_GATE_RES, "res": ops.EPI_BIAS_RES}[epi]
    if epi in ("gate", "res"):
        kw["res"] = hn("ir", (M, N), device=DEV)
    if epi == "gate":
        F_ = 3 if M % 3 == 0 else 1
        kw.update(e=hn("ie", (1, F_, 6, N), 0.5, device=DEV), mod=None, gate_idx=5, rows_per_batch=M, frame_len=M // F_)
    want = ops.gemm_w8a8(xq, sx, wq, sw, b, code, **kw)
    try:
        _set_tuning("gemm_asm", 19)
        got = ops.gemm_w8a8(xq, sx, wq, sw, b, code, **kw)
        again = ops.gemm_w8a8(xq, sx, wq, sw, b, code, **kw)
    finally:
        _set_tuning("gemm_asm", 35)
    assert torch.equal(got, again)
    if epi == "gelu":
        assert_bf16_close(got, want, 1, 0.995, f"w8a8 asm {M}x{N}x{K} {epi}")
    else:
        assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()


def test_synth_hash_kernel_matches_the_tensor_hash():
    """ll_synth_hash (the integer hash as one kernel of the library) against the int64 tensor evaluation it replaces on the GPU and
    against the CPU evaluation: bit-identical, any length (ragged last workgroup), both kinds."""
    try:
        for seed, name, shape in ((0, "blocks.3.ffn.0.weight", (257, 33)), (9, "noise", (5, 16, 7, 11)), (2, "x", (1,))):
            for fn in (synth.hash_uniform, synth.hash_normal):
                synth.FORCE_TORCH_HASH = False
                got = fn(seed, name, shape, device=DEV)
                synth.FORCE_TORCH_HASH = True
                ref_dev = fn(seed, name, shape, device=DEV)
                ref_cpu = fn(seed, name, shape)
                assert got.shape == ref_cpu.shape and torch.equal(got, ref_dev) and torch.equal(got.cpu(), ref_cpu), (seed, name, fn.__name__)
    finally:
        synth.FORCE_TORCH_HASH = False


@pytest.mark.parametrize("B,L,H,K,Sk", [(1, 1100, 12, 1536, 512),     # the 1.3B model's cross-attention shape class: 12 planes, 8 key tiles (M > 1024:
                                                                      # ops.gemm must take the plain kernel too, not the small-M split-K path)
                                        (2, 300, 2, 256, 512),        # batch 2, toy width (2 planes), padded rows + idle waves
                                        (1, 72, 3, 384, 640)])        # the last q-tile of Lq 4680; 3 planes; 10 key tiles
def test_cross_q_rmsnorm_fused_into_projection_and_attention(ops, B, L, H, K, Sk):
    """model.py:172,189 -- q = norm_q(self.q(x)); attention(q, k, v) -- as TWO launches (gemm_ssq: the projection leaves per-plane
    row sums of squares; flash_attn_qnorm: the attention kernel's Q prologue normalises) against the three-launch form
    (gemm, rmsnorm, flash_attn) and against the oracle's RMSNorm + exact attention."""
    C = H * 128
    x = hn("fx", (B, L, K))
    w = hn("fw", (C, K), 1.0 / math.sqrt(K))
    b = hn("fb", (C,), 0.1)
    nw = (1.0 + 0.1 * hn("fnw", (C,)).float()).to(bf)
    k = hn("fk", (B, Sk, H, 128))
    v = hn("fv", (B, Sk, H, 128), 0.7)
    assert ops.gemm_ssq_planes(B * L, C, K) == H and ops.flash_attn_qnorm_ok(H, Sk)
    xd, wd, bd, nwd, kd, vd = (t.to(DEV) for t in (x, w, b, nw, k, v))
    q3 = ops.gemm(xd, wd, bd)
    qraw, ssq = ops.gemm_ssq(xd, wd, bd)
    assert torch.equal(qraw, q3), "the SSQ epilogue must not change the projection's output"
    ref_ss = (q3.float().view(B * L, H, 128) ** 2).sum(-1).t()                     # [H, B*L]
    assert torch.allclose(ssq, ref_ss, rtol=2e-6, atol=0), (ssq - ref_ss).abs().max().item()
    qn = ops.rmsnorm(q3.view(B * L, C), nwd, 1e-6).view(B, L, H, 128)
    want = ops.flash_attn(qn, kd, vd, [(0, Sk)])
    got = ops.flash_attn_qnorm(qraw.view(B, L, H, 128), ssq, nwd, 1e-6, kd, vd, Sk)
    assert torch.isfinite(got.float()).all()
    # the same arithmetic with the same rounding points; rinv by v_rsq_f32 instead of 1 / sqrtf and another order of the row sum:
    # a bf16 flip of q in ~1e-4 of the elements, far below one output ulp after 128-long dot products
    assert (got.float() - want.float()).abs().max().item() < 4e-3, (got.float() - want.float()).abs().max().item()
    assert rel_l2(got.cpu(), want.cpu()) < 2e-3
    exact = R.attention_exact(R.rms_norm(q3.cpu().view(B * L, C), nw, 1e-6).view(B, L, H, 128), k, v)
    assert rel_l2(got.cpu(), exact) < 6e-3 and (got.cpu().double() - exact).abs().max().item() < 1.2e-2


def test_cross_q_fusion_entry_points_refuse_uncovered_shapes(ops):
    from longlive_amd import _lib
    lib = _lib.load()
    assert lib.ll_gemm_ssq_planes(300, 136, 256) == 0 and lib.ll_flash_attn_qnorm_ok(2, 16) == 0
    x = hn("rx", (64, 256)).to(DEV)
    w = hn("rw", (136, 256)).to(DEV)
    with pytest.raises(RuntimeError):
        ops.gemm_ssq(x, w, hn("rb", (136,)).to(DEV))


@pytest.mark.parametrize("K", [8, 1032, 2048, 2056, 8960, 9216, 9224])
def test_quantize_rows_every_kernel_form_equals_the_oracle(ops, K):
    """ll_quantize_rows keeps rows of up to 9216 elements in registers between the maximum and the rounding pass (4 or 18 chunks of
    16 bytes per lane) and falls back to the two-pass kernel beyond: all three forms give the oracle's codes and scales bit for bit,
    incl. ragged last chunks, an all-zero row and a row stride larger than K."""
    from oracle.ref_model import RefModel
    rows, ld = 37, K + 16
    x = hn(f"qr{K}", (rows, ld), 2.0)
    x[5] = 0
    x[11, : K] *= 40.0
    xd = x.to(DEV)
    q = torch.empty(rows, K, dtype=torch.int8, device=DEV)
    sc = torch.empty(rows, dtype=torch.float32, device=DEV)
    from longlive_amd import _lib
    _lib.check(_lib.load().ll_quantize_rows(xd.data_ptr(), q.data_ptr(), sc.data_ptr(), rows, K, ld, torch.cuda.current_stream().cuda_stream),
               "ll_quantize_rows")
    oq, osc = RefModel.quantize_rows(x[:, :K])
    assert torch.equal(q.cpu().double(), oq) and torch.equal(sc.cpu(), osc)
