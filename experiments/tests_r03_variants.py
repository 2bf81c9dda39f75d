"""NOT COLLECTED -- the GPU tests of the kernel variants that round 4 moved to experiments/ (attention_r03_variants.hip,
gemm_r02_variants.hip): stream-K attention, the 16x16x32 attention loop, the split-K GEMM hand-off.  Kept with the code they test."""

@pytest.fixture
def splitk_kernel():
    """The shipped tuning (gemm_asm = 3) sends split-K calls the generated 256 x 128 kernel covers to that kernel; the tests of the
    split-K kernel itself (gemm_kernel_v4sk: hand-off, epochs, fail-safe wait) against the HIP kernels switch the generated kernels
    off for their duration."""
    _set_tuning("gemm_asm", 0)
    yield
    _set_tuning("gemm_asm", 3)


@pytest.mark.parametrize("B,Lq,H,Sk,seg,W", [
    (2, 300, 3, 1500, (0, 1437), 5),      # 12 pairs x 23 tiles over 5 workgroups: whole pairs + head / tail parts, ragged keys
    (2, 300, 3, 1500, (0, 1437), 40),     # more workgroups than pairs: every pair cut into 3-4 parts (middle parts too)
    (1, 257, 2, 2000, (37, 1100), 3),     # key range that does not start at slot 0; padded waves in the last q-tile
    (1, 600, 1, 1100, (0, 1088), 8),      # 3 pairs x 17 tiles over 8 workgroups
    (1, 64, 1, 1024, (0, 1024), 16),      # one pair, one tile per workgroup
    (1, 520, 2, 1300, (0, 1300), 400),    # more workgroups than tile units: some workgroups have no work
])
def test_flash_attn_stream_k(ops, B, Lq, H, Sk, seg, W):
    """The stream-K cut (key-tile ranges over a fixed number of workgroups + log-sum-exp merge of the parts) at sizes the
    fp64 reference handles in full, with the workgroup count forced so that every kind of part occurs."""
    q = hn("aq", (B, Lq, H, 128))
    k = hn("ak", (B, Sk, H, 128))
    v = hn("av", (B, Sk, H, 128), 0.7)
    exact = R.attention_exact(q, k[:, seg[0]:seg[1]], v[:, seg[0]:seg[1]])
    try:
        _set_tuning("attn_sk_wgs", W)
        got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [seg]).cpu()
        _set_tuning("attn_sk_wgs", -1)
        base = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [seg]).cpu()
    finally:
        _set_tuning("attn_sk_wgs", -1)
    err, berr = (got.double() - exact).abs().max().item(), (base.double() - exact).abs().max().item()
    assert err < 1.2e-2 and rel_l2(got, exact) < 6e-3, (err, berr, rel_l2(got, exact))
    assert (got.float() - base.float()).abs().max().item() < 8e-3
    assert err < 2 * berr + 1e-3, f"stream-K max err {err} vs unsplit kernel {berr}"


def test_flash_attn_stream_k_rescale_across_parts(ops):
    """A spiked key in the LAST part of a split pair: the merge must rescale the earlier parts by 2^(c (m_i - M))."""
    B, Lq, H, Sk = 1, 64, 1, 2048
    q, k, v = hn("sq", (B, Lq, H, 128)), hn("sk", (B, Sk, H, 128)), hn("sv", (B, Sk, H, 128))
    k[0, 2040, 0] = (q[0, 5, 0].float() * 3).to(bf)      # huge score for query 5 in the last tile
    k[0, 3, 0] = (q[0, 9, 0].float() * 3).to(bf)         # and for query 9 in the first tile
    exact = R.attention_exact(q, k, v)
    try:
        _set_tuning("attn_sk_wgs", 4)
        got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [(0, Sk)]).cpu().double()
    finally:
        _set_tuning("attn_sk_wgs", -1)
    assert (got - exact).abs().max().item() < 2e-2




@pytest.mark.parametrize("B,Lq,H,Sk,seg", [
    (2, 300, 3, 1500, (0, 1437)),         # batch 2, ragged last key tile, padded waves in the last q-tile
    (1, 64, 1, 1024, (0, 1024)),          # exactly at the ping-pong threshold, all but two waves padding
    (1, 257, 2, 2000, (37, 1100)),        # key range that does not start at slot 0
    (1, 520, 12, 1300, (0, 1300)),        # real head count, 3 q-tiles
])
def test_flash_attn_mfma16_variant(ops, B, Lq, H, Sk, seg):
    """The ping-pong loop on v_mfma_f32_16x16x32_bf16 (tuning key attn_mfma16): same bar against fp64
    as the shipped kernel, and within two kernels' rounding of it element by element."""
    q = hn("aq", (B, Lq, H, 128))
    k = hn("ak", (B, Sk, H, 128))
    v = hn("av", (B, Sk, H, 128), 0.7)
    exact = R.attention_exact(q, k[:, seg[0]:seg[1]], v[:, seg[0]:seg[1]])
    try:
        _set_tuning("attn_asm", 0)
        base = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [seg]).cpu()
        _set_tuning("attn_mfma16", 1)
        got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [seg]).cpu()
    finally:
        _set_tuning("attn_mfma16", 0)
        _set_tuning("attn_asm", 1)
    err, berr = (got.double() - exact).abs().max().item(), (base.double() - exact).abs().max().item()
    assert err < 1.2e-2 and rel_l2(got, exact) < 6e-3, (err, berr, rel_l2(got, exact))
    assert (got.float() - base.float()).abs().max().item() < 8e-3


def test_flash_attn_mfma16_variant_rescale(ops):
    """Late and early running-max jumps (spiked keys in the last and the first tile of a 32-tile range) through the 16x16x32 loop:
    its running max is shared by the four lanes of a query and its row sum stays lane-partial until the epilogue."""
    B, Lq, H, Sk = 1, 64, 1, 2048
    q, k, v = hn("sq", (B, Lq, H, 128)), hn("sk", (B, Sk, H, 128)), hn("sv", (B, Sk, H, 128))
    k[0, 2040, 0] = (q[0, 5, 0].float() * 3).to(bf)
    k[0, 3, 0] = (q[0, 9, 0].float() * 3).to(bf)
    k[0, 1000, 0] = (q[0, 37, 0].float() * 3).to(bf)     # a query of the second 16-row block, mid-range
    exact = R.attention_exact(q, k, v)
    try:
        _set_tuning("attn_asm", 0)
        _set_tuning("attn_mfma16", 1)
        got = ops.flash_attn(q.to(DEV), k.to(DEV), v.to(DEV), [(0, Sk)]).cpu().double()
    finally:
        _set_tuning("attn_mfma16", 0)
        _set_tuning("attn_asm", 1)
    assert (got - exact).abs().max().item() < 2e-2




@pytest.mark.parametrize("M,N,K,epi", [(4680, 1536, 8960, "gate"), (4680, 1536, 8960, "bias"), (1560, 1536, 8960, "res"),
                                       (300, 512, 1024, "gelu"), (4680, 1536, 1536, "bias"), (300, 136, 1024, "bias")])
def test_gemm_splitk_matches_unsplit(ops, M, N, K, epi, splitk_kernel):
    """ll_gemm_bf16_splitk (256 x 256 tiles, K cut in two, halves exchanged through the workspace inside the kernel) against
    ll_gemm_bf16: same products, the fp32 sum split once more -> <= 1 bf16 ulp apart; 30 repeated launches are bit-identical
    (a stale or torn hand-off would show up as a run-to-run difference) and leave the workspace's error word at zero."""
    from longlive_amd import _lib
    x = hn("skx", (M, K), device=DEV)
    w = (hn("skw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("skb", (N,), 0.1, device=DEV)
    kw = {}
    code = {"bias": ops.EPI_BIAS, "gelu": ops.EPI_BIAS_GELU, "gate": ops.EPI_BIAS_GATE_RES, "res": ops.EPI_BIAS_RES}[epi]
    if epi in ("gate", "res"):
        kw["res"] = hn("skr", (M, N), device=DEV)
    if epi == "gate":
        F_ = 3
        kw.update(e=hn("ske", (1, F_, 6, N), 0.5, device=DEV), mod=hn("skm", (6, N), 0.1, device=DEV), gate_idx=5, rows_per_batch=M,
                  frame_len=M // F_)
    want = ops.gemm(x, w, b, code, **kw)
    eligible = _lib.load().ll_gemm_splitk_plan(M, N, K, 0) == 1
    assert eligible == (N % 256 == 0)
    first = ops.gemm(x, w, b, code, splitk=True, **kw)
    for _ in range(30):
        again = ops.gemm(x, w, b, code, splitk=True, **kw)
        assert torch.equal(again, first)
    torch.cuda.synchronize()
    ops.splitk_check()                                   # no hand-off timed out
    if eligible:
        ws = ops.splitk_workspace(x.device, M, N)
        assert int(ws[:4096].view(torch.int32)[-1]) == 0                  # error word clear; the flags hold launch epochs
        # a 1-ulp flip of bf16(acc + bias) (values up to ~4: ulp 2^-6) survives the gate / residual as an ABSOLUTE difference
        # while the sum itself may be small: absolute bound there, ulp bound for the plain epilogues
        fused = epi in ("gate", "res")
        assert_bf16_close(first, want, 2 if fused else 1, 0.99, f"splitk {M}x{N}x{K} {epi}", atol=4e-2 if fused else None)
    else:
        assert torch.equal(first, want)




def test_gemm_splitk_handoff_is_fresh_across_launches(ops, splitk_kernel):
    """The partial tiles live at fixed workspace addresses: alternate two different activations so that a partner reading the
    PREVIOUS launch's bytes (a stale line somewhere between the two workgroups) cannot reproduce the right answer."""
    M, N, K = 4680, 1536, 8960
    w = (hn("fw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("fb", (N,), 0.1, device=DEV)
    xs = [hn(f"fx{i}", (M, K), device=DEV) for i in range(2)]
    want = [ops.gemm(x, w, b, ops.EPI_BIAS, splitk=True) for x in xs]
    torch.cuda.synchronize()
    assert not torch.equal(want[0], want[1])
    for it in range(40):
        got = ops.gemm(xs[it & 1], w, b, ops.EPI_BIAS, splitk=True)
        assert torch.equal(got, want[it & 1]), f"launch {it}"
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                      # a second stream gets its own workspace; both run at once
        side.wait_stream(torch.cuda.current_stream())
        for it in range(10):
            got2 = ops.gemm(xs[1 - (it & 1)], w, b, ops.EPI_BIAS, splitk=True)
            assert torch.equal(got2, want[1 - (it & 1)])
    for it in range(10):
        got = ops.gemm(xs[it & 1], w, b, ops.EPI_BIAS, splitk=True)
        assert torch.equal(got, want[it & 1])
    torch.cuda.synchronize()




def test_gemm_splitk_handoff_is_fail_safe(ops, splitk_kernel):
    """The hand-off never hangs and never trusts a stale word: (1) a flag page full of garbage (what an aborted launch or a foreign
    writer could leave behind) changes nothing -- flags must equal THIS launch's epoch; (2) with the test hook that makes every
    second workgroup exit before it publishes (a partner that never arrives), the launch still completes within the bounded poll,
    ops.splitk_check() raises and names the workspace, and the next launch on the same workspace is correct again."""
    import time
    from longlive_amd import _lib
    lib = _lib.load()
    M, N, K = 4680, 1536, 8960
    w = (hn("zw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("zb", (N,), 0.1, device=DEV)
    x, x2 = hn("zx", (M, K), device=DEV), hn("zx2", (M, K), device=DEV)
    want = ops.gemm(x, w, b, ops.EPI_BIAS, splitk=True)
    ws = ops.splitk_workspace(x.device, M, N)
    ws[:4092].view(torch.int32).fill_(0x5a5a5a5a)        # every flag word poisoned; the error word (last of the page) stays 0
    for _ in range(3):
        assert torch.equal(ops.gemm(x, w, b, ops.EPI_BIAS, splitk=True), want)
    ops.splitk_check()
    try:
        assert lib.ll_set_tuning(b"gemm_splitk_fault", 1) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        broken = ops.gemm(x2, w, b, ops.EPI_BIAS, splitk=True)     # other input: the stale partner halves in the workspace are x's
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        lib.ll_set_tuning(b"gemm_splitk_fault", 0)
    assert 0.02 < dt < 2.0, dt                           # the 50 ms poll budget, not a hang
    with pytest.raises(RuntimeError, match="hand-off timed out"):
        ops.splitk_check()
    assert not torch.equal(broken, ops.gemm(x2, w, b, ops.EPI_BIAS))
    assert torch.equal(ops.gemm(x, w, b, ops.EPI_BIAS, splitk=True), want)
    ops.splitk_check()




def test_gemm_splitk_l2_exchange_matches(ops, splitk_kernel):
    """Opt-in exchange through the pair's L2 (tuning key gemm_splitk_l2; falls back to the sc1 form when the placement probe says
    partners do not share an XCD): same bits as the shipped exchange."""
    from longlive_amd import _lib
    M, N, K = 4680, 1536, 8960
    w = (hn("lw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("lb", (N,), 0.1, device=DEV)
    xs = [hn(f"lx{i}", (M, K), device=DEV) for i in range(2)]
    want = [ops.gemm(x, w, b, ops.EPI_BIAS, splitk=True) for x in xs]
    try:
        assert _lib.load().ll_set_tuning(b"gemm_splitk_l2", 1) == 0
        for it in range(20):
            assert torch.equal(ops.gemm(xs[it & 1], w, b, ops.EPI_BIAS, splitk=True), want[it & 1]), f"launch {it}"
    finally:
        _lib.load().ll_set_tuning(b"gemm_splitk_l2", 0)
    torch.cuda.synchronize()




def test_gemm_w8a8_splitk_is_exact(ops):
    """W8A8 split-K: the halves exchange int32 sums, so the result equals the unsplit kernel's bit for bit."""
    M, N, K = 4680, 1536, 8960
    x = hn("qx", (M, K), device=DEV)
    w = (hn("qw", (N, K), device=DEV) / math.sqrt(K)).to(bf)
    b = hn("qb", (N,), 0.1, device=DEV)
    xq, sx = ops.quantize_rows(x)
    wq, sw = ops.quantize_rows(w)
    res = hn("qr", (M, N), device=DEV)
    kw = dict(res=res, e=hn("qe", (1, 3, 6, N), 0.5, device=DEV), mod=hn("qm", (6, N), 0.1, device=DEV), gate_idx=5, rows_per_batch=M,
              frame_len=M // 3)
    want = ops.gemm_w8a8(xq, sx, wq, sw, b, ops.EPI_BIAS_GATE_RES, **kw)
    for _ in range(10):
        got = ops.gemm_w8a8(xq, sx, wq, sw, b, ops.EPI_BIAS_GATE_RES, splitk=True, **kw)
        assert torch.equal(got, want)
    torch.cuda.synchronize()


