"""The generated GEMM kernels (longlive_amd/csrc/gen/gemm_asm_gen.py) executed on the CPU by tools/gfx950_emu.py: one workgroup
(a 256 x WN tile with a ragged M edge, idle waves included) for every epilogue, against a numpy restatement of gemm_common.h's
rounding points; all three completion models of the emulator; lint clean; assembles for gfx950."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "longlive_amd", "csrc", "gen"))

import gemm_asm_gen as G          # noqa: E402
import gfx950_emu as E            # noqa: E402


def bf(x):
    return E.bf16_round(np.asarray(x, dtype=np.float32)).astype(np.uint16)


def f32(bits):
    return E.bf16_to_f32(np.asarray(bits, dtype=np.uint32))


def rbf(x):
    return f32(bf(x))


def run_case(WN, epi, mode, rows_valid=200, K=448, seed=0, m0=300, frame_len=130, row_lo=0, xpad=0, ypad=0):
    rng = np.random.default_rng(seed)
    text = G.generate(WN, epi, f"T{WN}E{epi}")
    assert G.lint(text) == []
    N = WN
    x = bf(rng.standard_normal((rows_valid, K)))
    w = bf(rng.standard_normal((N, K)) / np.sqrt(K))
    bias = bf(0.1 * rng.standard_normal(N))
    res = bf(rng.standard_normal((rows_valid, N)))
    nframes = (m0 + rows_valid - 1) // frame_len + 1      # exactly the frames the valid rows touch: rows past M must not index beyond them
    gate = bf(0.5 * rng.standard_normal((nframes, N)))
    mem = E.Memory()
    partial = epi == G.EPI_PARTIAL
    ssq = epi == G.EPI_BIAS_SSQ
    assq = mem.alloc(np.full(rows_valid, np.nan, dtype=np.float32)) if ssq else 0      # exactly the valid rows: a store past M raises
    # row strides larger than the row (ldx > K, ldo > N: column slices of wider tensors); the padding holds NaN patterns
    xs = np.full((rows_valid, K + xpad), 0x7FC0, dtype=np.uint16); xs[:, :K] = x
    rs = np.full((rows_valid, N + ypad), 0x7FC0, dtype=np.uint16); rs[:, :N] = res
    ax, aw, ab, ar, ag = mem.alloc(xs), mem.alloc(w), mem.alloc(bias), mem.alloc(rs), mem.alloc(gate)
    ay = mem.alloc(np.full((rows_valid, N + ypad), np.nan, dtype=np.float32) if partial else np.full((rows_valid, N + ypad), 0x7FC0, dtype=np.uint16))
    m = E.Machine(text, mem, 4, mode=mode, lds_bytes=G.Cfg(WN, epi).lds_bytes)
    for wv in m.waves:
        s = wv.s
        def put64(i, val):
            s[i], s[i + 1] = val & 0xFFFFFFFF, val >> 32
        put64(G.S_X, ax); put64(G.S_W, aw); put64(G.S_Y, ay); put64(G.S_BIAS, ab); put64(G.S_RES, ar); put64(G.S_GATE, ag)
        s[G.S_LDX], s[G.S_LDW], s[G.S_LDO] = (K + xpad) * 2, K * 2, (N + ypad) * (4 if partial else 2)
        s[G.S_ROWS], s[G.S_COLS], s[G.S_NK] = rows_valid, N, K // 64
        s[G.S_FLEN], s[G.S_GSTRIDE], s[G.S_M0] = frame_len, N * 2, m0
        s[G.S_ROWLO] = row_lo
        if ssq:
            put64(G.S_SSQ, assq)
        wv.v[G.V_TID] = 64 * wv.id + np.arange(64, dtype=np.uint32)
        wv.v[1:] = 0x7FC0BEEF
        wv.a[:] = 0x7FC0BEEF
    m.run()
    acc = (f32(x).astype(np.float64) @ f32(w).astype(np.float64).T).astype(np.float32)
    if ssq:
        run_case.last_ssq = mem.get(assq).view(np.float32).copy()
    if partial:                                       # one K-range of a split-K call: the fp32 accumulators as they stand
        yfull = mem.get(ay).view(np.float32).reshape(rows_valid, N + ypad)
        assert np.isnan(yfull[:, N:]).all(), "the kernel wrote past its N columns"
        return yfull[:, :N].astype(np.float64), acc.astype(np.float64)
    yfull = mem.get(ay).view(np.uint16).reshape(rows_valid, N + ypad)
    assert (yfull[:, N:] == 0x7FC0).all(), "the kernel wrote past its N columns"
    got = f32(yfull[:, :N]).astype(np.float64)
    v = rbf(acc + f32(bias)[None, :])
    if epi in (G.EPI_BIAS, G.EPI_BIAS_SSQ):
        want = v
    elif epi == G.EPI_GELU:
        xx = v.astype(np.float32)
        k0, k1, ce = np.float32(0.7978845608028654), np.float32(0.044715), np.float32(-2.0 * 1.4426950408889634)
        u = k0 * (xx + ((k1 * xx) * xx) * xx)
        e = np.exp2((ce * u).astype(np.float64)).astype(np.float32)
        want = rbf(xx * (np.float32(1.0) / (np.float32(1.0) + e)))
    elif epi == G.EPI_RES:
        want = rbf(f32(res) + v)
    else:
        frames = (m0 + np.arange(rows_valid)) // frame_len
        gt = f32(gate)[frames]
        want = rbf(f32(res) + rbf(v * gt))
    return got, want.astype(np.float64)


@pytest.mark.parametrize("epi", [G.EPI_BIAS, G.EPI_GELU, G.EPI_GATE_RES, G.EPI_RES])
def test_gemm_asm_epilogues(epi):
    got, want = run_case(224, epi, "lazy")
    assert np.isfinite(got).all()
    # one bf16 ulp of the fp32 accumulation-order difference at most, almost all elements identical
    ulp = np.maximum(np.abs(want), 2.0 ** -6) * 2.0 ** -7
    assert (np.abs(got - want) <= (1.01 if epi == G.EPI_BIAS else 2.02) * ulp + (0 if epi in (G.EPI_BIAS, G.EPI_GELU) else 4e-2)).all(), np.abs(got - want).max()
    assert (got == want).mean() > 0.97, (got == want).mean()


def test_gemm_asm_eager_model_and_small_tile():
    got, want = run_case(128, G.EPI_BIAS, "eager", rows_valid=70, K=832)
    assert (got == want).mean() > 0.97 and np.abs(got - want).max() < 0.05


def test_gemm_asm_mixed_model_catches_no_early_refill():
    """LDS-DMA lands at once, LDS reads only when waited for: a ring slot / unit refilled before its fragment reads have been
    retired would show here (the X units are wave-private and guarded by lgkmcnt order alone, not by a barrier)."""
    got, want = run_case(224, G.EPI_BIAS, "mixed", rows_valid=256, K=1024)
    assert (got == want).mean() > 0.97 and np.abs(got - want).max() < 0.05


def test_gemm_asm_widest_tile_fills_the_lds():
    assert G.Cfg(256, G.EPI_RES).lds_bytes == 160 * 1024
    got, want = run_case(256, G.EPI_RES, "lazy", rows_valid=130, K=512)
    assert (got == want).mean() > 0.97 and np.abs(got - want).max() < 0.05


def test_gemm_asm_qkv_tile_stores_only_its_row_window():
    """The 192-wide kernel of the fused QKV projection: a V tile stores rows row_lo <= row < rows only (tokens outside the cache
    insert window are not written: gemm_common.h epi_dest)."""
    got, want = run_case(192, G.EPI_BIAS, "lazy", rows_valid=150, K=320, row_lo=37)
    assert np.isnan(got[:37]).all(), "rows below the window were written"
    assert (got[37:] == want[37:]).mean() > 0.97 and np.abs(got[37:] - want[37:]).max() < 0.05


@pytest.mark.parametrize("rows_valid,K", [(1, 256), (65, 256), (256, 320)])
def test_gemm_asm_shortest_k_and_single_row(rows_valid, K):
    """K = 256 is the shortest the launcher sends here (4 K-steps: fewer than the loop's unroll, the prologue stages past the end
    and re-loads the last step); one valid row leaves three idle waves and 63 masked lanes."""
    got, want = run_case(128, G.EPI_GATE_RES, "lazy", rows_valid=rows_valid, K=K, m0=0, frame_len=40)
    assert np.isfinite(got).all()
    assert (got == want).mean() > 0.95 and np.abs(got - want).max() < 0.07


def test_gemm_asm_partial_sums_for_split_k():
    """EPI_PARTIAL: the kernel of the small-M split-K path stores its fp32 accumulators (rows past M untouched)."""
    got, want = run_case(128, G.EPI_PARTIAL, "lazy", rows_valid=100, K=512)
    assert np.isfinite(got).all() and np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("rows_valid,mode", [(200, "lazy"), (256, "mixed"), (33, "eager")])
def test_gemm_asm_row_sums_of_squares(rows_valid, mode):
    """EPI_BIAS_SSQ: the bias epilogue + ssq[row] = sum over the tile's 128 columns of the ROUNDED outputs squared (the statistics
    of the RMSNorm that follows the projection: model.py:78-86 after :172).  The outputs equal the bias kernel's; the sums are the
    fp32 sums of exactly those bf16 values (any order: compared to an fp64 sum), rows past M are never written."""
    got, want = run_case(128, G.EPI_BIAS_SSQ, mode, rows_valid=rows_valid, K=448)
    plain, _ = run_case(128, G.EPI_BIAS, mode, rows_valid=rows_valid, K=448)
    assert np.array_equal(got, plain)
    ss = run_case.last_ssq.astype(np.float64)
    ref = (got ** 2).sum(axis=1)
    assert np.isfinite(ss).all() and np.abs(ss - ref).max() <= 2e-6 * ref.max(), np.abs(ss - ref).max()


def test_gemm_asm_row_strides_wider_than_the_rows():
    """ldx > K and ldo > N (the ABI lets X / Y / RES be column slices of wider tensors): nothing outside the slice is read into
    the result or written."""
    for epi in (G.EPI_GATE_RES, G.EPI_GELU):
        got, want = run_case(128 if epi == G.EPI_GATE_RES else 224, epi, "lazy", rows_valid=150, K=320, xpad=24, ypad=40)
        assert np.isfinite(got).all() and (got == want).mean() > 0.95 and np.abs(got - want).max() < 0.07


def test_gemm_asm_random_geometries():
    """Seeded sweep over tile widths, epilogues, ragged row counts, K lengths, frame lengths and row windows (the gate-row read past
    M of round 3 was a geometry nobody had written down: small frames under a ragged last tile)."""
    rng = np.random.default_rng(20261004)
    for _ in range(8):
        WN = int(rng.choice([128, 128, 192, 224, 256]))
        epi = int(rng.choice([G.EPI_BIAS, G.EPI_GELU, G.EPI_GATE_RES, G.EPI_RES, G.EPI_PARTIAL]))
        rows = int(rng.integers(1, 257))
        K = 64 * int(rng.integers(4, 10))
        flen = int(rng.integers(1, 200))
        m0 = int(rng.integers(0, 5)) * 256
        lo = int(rng.integers(0, rows)) if (epi == G.EPI_BIAS and rng.random() < 0.5) else 0
        mode = str(rng.choice(["lazy", "eager", "mixed"]))
        got, want = run_case(WN, epi, mode, rows_valid=rows, K=K, seed=int(rng.integers(1 << 30)), m0=m0, frame_len=flen, row_lo=lo)
        tag = (WN, epi, rows, K, flen, m0, lo, mode)
        if lo:
            assert np.isnan(got[:lo]).all(), tag
        g, w = got[lo:], want[lo:]
        assert np.isfinite(g).all(), tag
        if epi == G.EPI_PARTIAL:
            assert np.abs(g - w).max() < 2e-5 * max(1.0, np.abs(w).max()), tag
        else:
            assert (g == w).mean() > 0.93 and np.abs(g - w).max() < 0.13, (tag, (g == w).mean(), np.abs(g - w).max())


def run_case_i8(WN, epi, mode, rows_valid=150, K=512, seed=0, m0=256, frame_len=90):
    """W8A8 form: int8 operands, exact int32 sums, v = bf16(float(acc) * (sx[m] * sw[n]) + bias) (gemm_common.h), same epilogues."""
    rng = np.random.default_rng(seed)
    text = G.generate(WN, epi, f"Q{WN}E{epi}", True)
    assert G.lint(text) == []
    N = WN
    x = rng.integers(-127, 128, (rows_valid, K)).astype(np.int8)
    w = rng.integers(-127, 128, (N, K)).astype(np.int8)
    sx = (0.5 + rng.random(rows_valid)).astype(np.float32) / np.float32(127 * np.sqrt(K))
    sw = (0.5 + rng.random(N)).astype(np.float32) / np.float32(40)
    bias = bf(0.1 * rng.standard_normal(N))
    res = bf(rng.standard_normal((rows_valid, N)))
    nframes = (m0 + rows_valid - 1) // frame_len + 1
    gate = bf(0.5 * rng.standard_normal((nframes, N)))
    mem = E.Memory()
    ax, aw, ab, ar, ag, asx, asw = (mem.alloc(t) for t in (x, w, bias, res, gate, sx, sw))
    ay = mem.alloc(np.full((rows_valid, N), 0x7FC0, dtype=np.uint16))
    m = E.Machine(text, mem, 4, mode=mode, lds_bytes=G.Cfg(WN, epi, True).lds_bytes)
    for wv in m.waves:
        s = wv.s
        def put64(i, val):
            s[i], s[i + 1] = val & 0xFFFFFFFF, val >> 32
        put64(G.S_X, ax); put64(G.S_W, aw); put64(G.S_Y, ay); put64(G.S_BIAS, ab); put64(G.S_RES, ar); put64(G.S_GATE, ag)
        put64(G.S_SX, asx); put64(G.S_SW, asw)
        s[G.S_LDX], s[G.S_LDW], s[G.S_LDO] = K, K, N * 2
        s[G.S_ROWS], s[G.S_COLS], s[G.S_NK] = rows_valid, N, K // 128
        s[G.S_FLEN], s[G.S_GSTRIDE], s[G.S_M0] = frame_len, N * 2, m0
        s[G.S_ROWLO] = 0
        wv.v[G.V_TID] = 64 * wv.id + np.arange(64, dtype=np.uint32)
        wv.v[1:] = 0x7FC0BEEF
        wv.a[:] = 0x7FC0BEEF
    m.run()
    got = f32(mem.get(ay).view(np.uint16).reshape(rows_valid, N)).astype(np.float64)
    acc = (x.astype(np.int64) @ w.astype(np.int64).T).astype(np.float32)
    scl = (sx[:, None] * sw[None, :]).astype(np.float32)
    v = rbf((acc * scl).astype(np.float32) + f32(bias)[None, :])
    if epi == G.EPI_BIAS:
        want = v
    elif epi == G.EPI_GELU:
        xx = v.astype(np.float32)
        k0, k1, ce = np.float32(0.7978845608028654), np.float32(0.044715), np.float32(-2.0 * 1.4426950408889634)
        u = k0 * (xx + ((k1 * xx) * xx) * xx)
        e = np.exp2((ce * u).astype(np.float64)).astype(np.float32)
        want = rbf(xx * (np.float32(1.0) / (np.float32(1.0) + e)))
    elif epi == G.EPI_RES:
        want = rbf(f32(res) + v)
    else:
        frames = (m0 + np.arange(rows_valid)) // frame_len
        want = rbf(f32(res) + rbf(v * f32(gate)[frames]))
    return got, want.astype(np.float64)


@pytest.mark.parametrize("WN,epi,mode", [(128, G.EPI_BIAS, "lazy"), (128, G.EPI_GATE_RES, "lazy"), (128, G.EPI_RES, "mixed"),
                                         (192, G.EPI_BIAS, "eager"), (224, G.EPI_GELU, "lazy")])
def test_gemm_asm_w8a8(WN, epi, mode):
    """The W8A8 variant (v_mfma_i32_32x32x32_i8, scales applied in the epilogue): integer sums are exact, so the bias epilogue
    must reproduce the rounding-point reference bit for bit; the others within the bf16 kernels' bounds."""
    got, want = run_case_i8(WN, epi, mode)
    assert np.isfinite(got).all()
    if epi == G.EPI_BIAS:
        assert (got == want).all(), np.abs(got - want).max()
    else:
        assert (got == want).mean() > 0.97 and np.abs(got - want).max() < 0.07, ((got == want).mean(), np.abs(got - want).max())


def test_gemm_asm_text_assembles(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    for epi, i8 in [(e, False) for e in (G.EPI_BIAS, G.EPI_GELU, G.EPI_GATE_RES, G.EPI_RES, G.EPI_PARTIAL)] + [(G.EPI_GELU, True), (G.EPI_GATE_RES, True)]:
        src = tmp_path / f"k{epi}.s"
        src.write_text('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n.text\nkernel:\n' + G.generate(224, epi, f"A{epi}", i8))
        r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "k.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[:2000]


# ---- persistent form: a workgroup walks several tiles and stages the next tile's first pieces under the current epilogue ---------
def run_persistent(WN, epi, mode, M=600, ntn=2, K=448, grid=2, gm=2, seed=0, frame_len=130, vcache=None):
    """The whole launch, one emulated workgroup after the other over shared memory: Y[M, ntn * WN] = epilogue(X W^T + b).  vcache =
    (S, v_lo, v_hi, v_shift): the last third of the columns is the V third of a fused QKV projection, redirected into cache_v [S, C]."""
    rng = np.random.default_rng(seed)
    text = G.generate(WN, epi, f"P{WN}E{epi}", False, True)
    assert G.lint(text) == []
    N = ntn * WN
    ntm = (M + 255) // 256
    x = bf(rng.standard_normal((M, K)))
    w = bf(rng.standard_normal((N, K)) / np.sqrt(K))
    bias = bf(0.1 * rng.standard_normal(N))
    res = bf(rng.standard_normal((M, N)))
    nframes = (M - 1) // frame_len + 1
    gate = bf(0.5 * rng.standard_normal((nframes, N)))
    mem = E.Memory()
    ax, aw, ab, ar, ag = mem.alloc(x), mem.alloc(w), mem.alloc(bias), mem.alloc(res), mem.alloc(gate)
    ay = mem.alloc(np.full((M, N), 0x7FC0, dtype=np.uint16))
    av = 0
    if vcache is not None:
        S, v_lo, v_hi, v_shift = vcache
        C = N // 3
        av = mem.alloc(np.full((S, C), 0x7FC0, dtype=np.uint16))
    ntiles = ntm * ntn
    for wg in range(min(grid, ntiles)):
        m = E.Machine(text, mem, 4, mode=mode, lds_bytes=G.Cfg(WN, epi).lds_bytes)
        for wv in m.waves:
            s = wv.s
            def put64(i, val):
                s[i], s[i + 1] = val & 0xFFFFFFFF, val >> 32
            put64(G.S_XB, ax); put64(G.S_WB, aw); put64(G.S_YB, ay); put64(G.S_BIASB, ab); put64(G.S_RESB, ar); put64(G.S_GATEB, ag)
            s[G.S_LDX], s[G.S_LDW], s[G.S_LDO0] = K * 2, K * 2, N * 2
            s[G.S_MM], s[G.S_NN], s[G.S_NK] = M, N, K // 64
            s[G.S_FLEN], s[G.S_GSTRIDE] = frame_len, N * 2
            s[G.S_TILE], s[G.S_GRID], s[G.S_NTILES], s[G.S_NTM], s[G.S_NTN], s[G.S_GM] = wg, min(grid, ntiles), ntiles, ntm, ntn, gm
            put64(G.S_VOUT, av)
            if vcache is not None:
                s[G.S_VCOL0], s[G.S_VC2], s[G.S_VSHIFT], s[G.S_VLO], s[G.S_VHI] = 2 * C, C * 2, v_shift & 0xFFFFFFFF, v_lo, v_hi
            wv.v[G.V_TID] = 64 * wv.id + np.arange(64, dtype=np.uint32)
            wv.v[1:] = 0x7FC0BEEF
            wv.a[:] = 0x7FC0BEEF
        m.run()
    acc = (f32(x).astype(np.float64) @ f32(w).astype(np.float64).T).astype(np.float32)
    v = rbf(acc + f32(bias)[None, :])
    if epi == G.EPI_BIAS:
        want = v
    elif epi == G.EPI_GELU:
        xx = v.astype(np.float32)
        k0, k1, ce = np.float32(0.7978845608028654), np.float32(0.044715), np.float32(-2.0 * 1.4426950408889634)
        u = k0 * (xx + ((k1 * xx) * xx) * xx)
        e = np.exp2((ce * u).astype(np.float64)).astype(np.float32)
        want = rbf(xx * (np.float32(1.0) / (np.float32(1.0) + e)))
    elif epi == G.EPI_RES:
        want = rbf(f32(res) + v)
    else:
        want = rbf(f32(res) + rbf(v * f32(gate)[np.arange(M) // frame_len]))
    yraw = mem.get(ay).view(np.uint16).reshape(M, N)
    got = f32(yraw).astype(np.float64)
    cache = None if vcache is None else mem.get(av).view(np.uint16).reshape(vcache[0], N // 3)
    return got, want.astype(np.float64), yraw, cache


@pytest.mark.parametrize("WN,epi,mode,grid,gm", [(224, G.EPI_GELU, "lazy", 2, 2), (128, G.EPI_GATE_RES, "mixed", 4, 4), (192, G.EPI_BIAS, "eager", 3, 1),
                                                 (128, G.EPI_RES, "lazy", 8, 4)])
def test_gemm_asm_persistent_walks_every_tile(WN, epi, mode, grid, gm):
    """generate(persistent=True): workgroup w computes tiles w, w + grid, ... of the launch (the tile -> (m-tile, n-tile) map of
    gemm_common.h evaluated in the kernel), staging the next tile's W(0..2) / X(0..1) before the current epilogue.  Every element of
    Y is written exactly once with the classic kernel's arithmetic: ragged last m-tile (88 rows: waves that are idle in one tile
    and active in the next), more tiles than workgroups, a workgroup count that does not divide them."""
    got, want, yraw, _ = run_persistent(WN, epi, mode, M=600, ntn=2, K=448, grid=grid, gm=gm)
    assert not (yraw == 0x7FC0).all(axis=1).any(), "a tile was never written"
    assert np.isfinite(got).all()
    assert (got == want).mean() > 0.96 and np.abs(got - want).max() < 0.08, ((got == want).mean(), np.abs(got - want).max())


def test_gemm_asm_persistent_equals_the_classic_kernel_bit_for_bit():
    """Same tile, same K order, same epilogue text: the persistent form's output bits are the classic form's."""
    got_p, _, _, _ = run_persistent(128, G.EPI_GATE_RES, "lazy", M=256, ntn=1, K=448, grid=1, gm=1, seed=7, frame_len=130)
    rng_case = run_case(128, G.EPI_GATE_RES, "lazy", rows_valid=256, K=448, seed=7, m0=0, frame_len=130)
    assert np.array_equal(got_p, rng_case[0])


@pytest.mark.parametrize("v_lo,v_hi,grid", [(100, 560, 2), (300, 600, 5), (0, 200, 3)])
def test_gemm_asm_persistent_qkv_v_redirect(v_lo, v_hi, grid):
    """The fused QKV projection on the persistent 192-wide kernel: tiles of the V third (columns >= 2 C) store token t into cache row
    t + v_shift for v_lo <= t < v_hi only -- per-tile base, row stride and row window computed in the kernel; V tiles with nothing
    to store are skipped (and never prefetched)."""
    M, C, S, v_shift = 600, 192, 900, 250
    got, want, yraw, cache = run_persistent(192, G.EPI_BIAS, "lazy", M=M, ntn=3, K=320, grid=grid, gm=2, seed=3, vcache=(S, v_lo, v_hi, v_shift))
    qk = slice(0, 2 * C)
    assert (got[:, qk] == want[:, qk]).mean() > 0.97 and np.abs(got[:, qk] - want[:, qk]).max() < 0.05
    assert (yraw[:, 2 * C:] == 0x7FC0).all(), "the V third of Y must stay unwritten"
    cv = f32(cache).astype(np.float64)
    hi = min(M, v_hi)
    rows = np.arange(v_lo, hi) + v_shift
    assert (cv[rows] == want[v_lo:hi, 2 * C:]).mean() > 0.97 and np.abs(cv[rows] - want[v_lo:hi, 2 * C:]).max() < 0.05
    other = np.ones(S, dtype=bool); other[rows] = False
    assert (cache[other] == 0x7FC0).all(), "cache rows outside the insert window were written"
