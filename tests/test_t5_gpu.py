"""Parity of the umT5 encoder kernels and of the assembled encoder (longlive_amd/text_encoder.py) against the CPU oracle
(oracle/ref_t5.py, pinned bit-exact to the reference's T5Encoder) and the goldens the reference itself produced
(tests/golden/t5_enc.pt).  Through the C ABI."""
import os

import pytest
import torch

from conftest import GOLDEN, load_golden
from longlive_amd import synth
from util import assert_bf16_close, bf, rel_l2, cosine

pytestmark = pytest.mark.gpu
DEV = "cuda"
TINY = dict(vocab_size=512, dim=256, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=3, text_len=64)


@pytest.fixture(scope="module")
def ops():
    from longlive_amd import ops as o
    return o


def hn(name, shape, scale=1.0, shift=0.0, seed=91):
    return (synth.hash_normal(seed, name, shape) * scale + shift).to(bf)


@pytest.mark.parametrize("rows,C", [(70, 4096), (5, 256), (3, 520)])
def test_t5_rmsnorm(ops, rows, C):
    from oracle import ref_t5 as RT
    x, w = hn("x", (rows, C), 2.0, 0.1), hn("w", (C,), 0.1, 1.0)
    assert_bf16_close(ops.t5_rmsnorm(x.to(DEV), w.to(DEV)), RT.t5_layer_norm(x, w), 1, 0.99, "t5_rmsnorm")


def test_t5_gated_gelu(ops):
    from oracle import ref_t5 as RT
    h = hn("h", (37, 2 * 512), 1.5)
    want = h[:, 512:] * RT.gelu_py(h[:, :512])
    got = ops.t5_gated_gelu(h.to(DEV))
    # bf16(1 + bf16(tanh)) is ill-conditioned near tanh = -1: a last-bit difference between the CPU's and the GPU's tanhf
    # flips 1 - 0.99609375 to 1 - 0.9921875 (a factor 2 on a value ~0.01), so: > 98 % bit-exact, the rest small in absolute terms
    assert_bf16_close(got, want, 2, 0.98, "t5_gated_gelu", atol=2e-2)
    assert rel_l2(got.cpu(), want) < 5e-3


def test_gather_rows(ops):
    table = hn("tab", (300, 256))
    ids = torch.tensor([0, 299, 7, 7, 123], dtype=torch.long)
    assert torch.equal(ops.gather_rows(table.to(DEV), ids.to(DEV)).cpu(), table[ids])


@pytest.mark.parametrize("L,H,n", [(64, 4, 23), (128, 2, 128), (512, 64, 77), (512, 3, 512)])
def test_t5_attention(ops, L, H, n):
    """One head-batch of T5Attention's core (t5.py:97-111) against the same torch ops on the CPU."""
    from longlive_amd.text_encoder import relative_bucket_table
    C = H * 64
    q, k, v = hn("q", (L, C), 0.35), hn("k", (L, C)), hn("v", (L, C))
    emb = hn("emb", (32, H))
    tab = emb[relative_bucket_table(L)].t().contiguous()                            # [H, 2L-1]
    idx = torch.arange(L)[None, :] - torch.arange(L)[:, None] + L - 1               # [Lq, Lk]
    bias = tab[:, idx]                                                              # [H, Lq, Lk]
    mask = torch.zeros(L, dtype=torch.long); mask[:n] = 1
    attn_bias = torch.zeros(1, H, L, L, dtype=bf) + bias[None]
    attn_bias.masked_fill_(mask.view(1, 1, 1, -1) == 0, torch.finfo(bf).min)
    qh, kh, vh = (t.view(1, L, H, 64) for t in (q, k, v))
    attn = torch.einsum("binc,bjnc->bnij", qh, kh) + attn_bias
    attn = torch.softmax(attn.float(), dim=-1).type_as(attn)
    want = torch.einsum("bnij,bjnc->binc", attn, vh).reshape(L, C)
    qk = torch.cat([q, k], 1).contiguous().to(DEV)
    got = ops.t5_attention(qk, v.t().contiguous().to(DEV), tab.to(DEV), H, n)
    assert_bf16_close(got, want, 2, 0.95, f"t5_attention L={L} H={H} n={n}", atol=6e-3)   # cancellation near 0


def _encoder(cfg, seed=7):
    from longlive_amd.text_encoder import WanTextEncoder
    m = WanTextEncoder(cfg, device=DEV)
    m.load_state_dict(synth.synth_t5_state_dict(cfg, seed=seed, device=DEV))
    return m


def test_t5_encoder_tiny_vs_reference_golden():
    rec = load_golden("t5_enc.pt")["tiny"]
    cfg = synth.T5Config(**TINY)
    ids, mask = synth.synth_token_ids(cfg, rec["ntok"], seed=3, batch=2)
    got = _encoder(cfg).encode_ids(ids, mask)["prompt_embeds"].cpu()
    want = rec["out"]
    assert got.shape == want.shape and got.dtype == bf
    assert torch.all(got[0, 23:] == 0) and torch.all(got[1, 20:] == 0)
    r = rel_l2(got, want)
    # tolerance = the reference's own bf16 noise: its bf16 output is 1.7e-2 from an fp32 evaluation of the same weights
    # (this 3-layer toy amplifies a 1e-3 per-layer perturbation ~5x per layer; teacher-forced per-layer error is <= 1e-3)
    from oracle import ref_t5 as RT
    sd32 = {k: v.float() for k, v in synth.synth_t5_state_dict(cfg, seed=7).items()}
    floor = rel_l2(want, RT.text_encoder_forward(ids, mask, sd32, cfg.num_layers, cfg.num_heads))
    print(f"t5 tiny rel-L2 {r:.2e} (reference bf16 vs fp32: {floor:.2e})")
    assert r < floor and r < 2e-2 and cosine(got, want) > 0.9998


def test_t5_encoder_real_width_vs_reference_golden():
    """dim 4096, 64 heads x 64, ffn 10240, 512 positions, 2 layers against the reference T5Encoder's bf16 CPU output.
    Tolerance rel-L2 2e-2, the scale of the reference's own bf16-vs-fp32 noise (see the tiny test)."""
    rec = load_golden("t5_enc.pt")["wide"]
    cfg = synth.T5Config(vocab_size=4096, num_layers=2)
    ids, mask = synth.synth_token_ids(cfg, rec["ntok"], seed=3)
    got = _encoder(cfg).encode_ids(ids, mask)["prompt_embeds"].cpu()
    want = rec["out"]
    r = rel_l2(got, want)
    print(f"t5 real-width rel-L2 {r:.2e}")
    assert got.shape == (1, 512, 4096) and r < 2e-2 and cosine(got, want) > 0.9998


def test_t5_encoder_full_depth_vs_reference_golden():
    """The encoder at its FULL depth -- 24 layers, dim 4096, 64 heads x 64, ffn 10240, 512 positions -- against the reference
    T5Encoder's own bf16 CPU run (tests/golden/t5_enc_deep.pt, oracle/make_golden.py t5_deep: final context + the residual stream
    of the valid rows after layers 6 / 12 / 18 / 24).
    (a) teacher-forced: every 6-layer segment entered with the REFERENCE's hidden state must land on the reference's next state
        (rel-L2 < 3e-2, measured 2.4e-2 at worst; the 2-layer run measures 7.6e-3); a wrong layer, weight or bias table anywhere in the stack fails here;
    (b) free-running over all 24 layers the two implementations' bf16 rounding differences are amplified ~5 % per layer by this
        random-weight stack (1.0e-1 measured, cosine 0.9945): held to 1.5e-1 / 0.99, finite, padding rows zero."""
    if not os.path.exists(os.path.join(GOLDEN, "t5_enc_deep.pt")):
        pytest.skip("golden missing")
    rec = load_golden("t5_enc_deep.pt")
    n = rec["ntok"]
    cfg = synth.T5Config(vocab_size=4096, num_layers=24)
    ids, mask = synth.synth_token_ids(cfg, n, seed=3)
    enc = _encoder(cfg)
    states = {0: rec["x_in"], **rec["x_after"]}
    worst = 0.0
    for first in (0, 6, 12, 18):
        x = torch.zeros(cfg.text_len, cfg.dim, dtype=bf, device=DEV)
        x[:n] = states[first].to(DEV)
        xo, h = enc.text_encoder.run_layers(x, n, first, first + 6)
        r = rel_l2(xo[:n].cpu(), states[first + 6])
        worst = max(worst, r)
        assert r < 3e-2, (first, r)
        if first == 18:                                       # the encoder's final norm follows layer 23
            rf = rel_l2(h[:n].cpu(), rec["out"][0, :n])
            assert rf < 3e-2, rf
    got = enc.encode_ids(ids, mask)["prompt_embeds"].cpu()
    want = rec["out"]
    r = rel_l2(got, want)
    print(f"t5 full depth (24 layers): teacher-forced 6-layer segments worst rel-L2 {worst:.2e}; free-running rel-L2 {r:.2e}, cosine {cosine(got, want):.6f}")
    assert got.shape == (1, 512, 4096) and torch.isfinite(got.float()).all()
    assert r < 1.5e-1 and cosine(got, want) > 0.99, r
    assert (got[0, n:] == 0).all()


def test_text_encoder_rejects_bad_inputs():
    cfg = synth.T5Config(**TINY)
    m = _encoder(cfg)
    ids, mask = synth.synth_token_ids(cfg, 10, seed=3)
    with pytest.raises(RuntimeError):
        m.encode_ids(ids[:, :32], mask[:, :32])                   # not padded to text_len
    bad = ids.clone(); bad[0, 0] = cfg.vocab_size
    with pytest.raises(RuntimeError):
        m.encode_ids(bad, mask)
    with pytest.raises(RuntimeError):
        m(["a prompt"])                                           # no tokenizer injected
