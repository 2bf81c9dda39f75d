"""The generated self-attention kernel (longlive_amd/csrc/gen/attn_asm_gen.py) executed on the CPU by tools/gfx950_emu.py --
the text that is assembled into the library, one workgroup, against an fp64 attention.  Both completion models of the emulator
must pass: `lazy` (a memory operation completes only when an s_waitcnt retires it: missing / under-counted waits show as NaN) and
`eager` (it completes at issue: a ring slot overwritten too early shows as wrong scores).  Also: the hazard linter is clean and the
text assembles for gfx950."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "longlive_amd", "csrc", "gen"))

import attn_asm_gen as G          # noqa: E402
import gfx950_emu as E            # noqa: E402


@pytest.fixture(scope="module")
def kernel_text():
    return G.generate("buffer", "LLB")


def bf16_bits(x):
    return E.bf16_round(np.asarray(x, dtype=np.float32)).astype(np.uint16)


def bf16_val(bits):
    return E.bf16_to_f32(np.asarray(bits, dtype=np.uint32))


def run_case(text, mode, rows_valid, nkeys, seed=0, spikes=(), nheads=2, head=1, kstart=3, qnorm=False, eps=1e-6):
    """One workgroup: 256 query-row slots of which rows_valid exist, keys [kstart, kstart + nkeys) of a cache with nheads heads.
    qnorm: `text` is the QNORM form -- q is the raw projection output, RMS-normalised over all nheads * 128 channels in the kernel's
    prologue from per-(128-column plane, row) sums of squares (what gemm_asm_128_bias_ssq leaves), times a bf16 weight."""
    rng = np.random.default_rng(seed)
    D = 128
    ldq = ldo = ldk = nheads * D
    nq = rows_valid
    q = rng.standard_normal((nq, nheads, D)).astype(np.float32)
    if qnorm:
        q = (q * (0.5 + 2.0 * rng.random((nq, 1, 1)))).astype(np.float32)           # rows of different norms
        qraw = bf16_val(bf16_bits(q)).astype(np.float32)
        planes = (qraw.astype(np.float64) ** 2).sum(axis=2).T.astype(np.float32).copy()      # [nheads planes][nq]: exactly the valid rows
        wn = bf16_val(bf16_bits(1.0 + 0.1 * rng.standard_normal((nheads, D)))).astype(np.float32)
        ss = np.zeros(nq, dtype=np.float32)
        for j in range(nheads):                       # plane order, fp32, as the kernel sums
            ss = (ss + planes[j]).astype(np.float32)
        rinv = (1.0 / np.sqrt((ss.astype(np.float64) * np.float64(np.float32(1.0 / (nheads * D)))).astype(np.float32).astype(np.float64)
                              + np.float64(np.float32(eps)))).astype(np.float32)
        t1 = bf16_val(bf16_bits(qraw * rinv[:, None, None])).astype(np.float32)
        q_eff = bf16_val(bf16_bits(t1 * wn[None])).astype(np.float32)                # what the rmsnorm kernel would have written
    qsrc = q
    cache_rows = kstart + nkeys                      # the cache ENDS at the last key: anything past it is out of bounds
    k = rng.standard_normal((cache_rows, nheads, D)).astype(np.float32)
    v = (0.7 * rng.standard_normal((cache_rows, nheads, D))).astype(np.float32)
    for (qi, ki, amp) in spikes:
        k[kstart + ki, head] = amp * q[qi, head]
    if qnorm:
        for (qi, ki, amp) in spikes:                  # spikes relative to the NORMALISED query
            k[kstart + ki, head] = amp * q_eff[qi, head]
    qb_, kb_, vb_ = bf16_bits(q), bf16_bits(k), bf16_bits(v)
    mem = E.Memory()
    aq, ak, av = mem.alloc(qb_), mem.alloc(kb_), mem.alloc(vb_)
    if qnorm:
        assq, anw = mem.alloc(planes), mem.alloc(bf16_bits(wn))
    ao = mem.alloc(np.full((nq, nheads, D), 0x7FC0, dtype=np.uint16))
    m = E.Machine(text, mem, 4, mode=mode)
    scale = 1.0 / math.sqrt(D)
    c = np.float32(scale * 1.4426950408889634)
    nt = (nkeys + 63) // 64
    lastv = nkeys - 64 * (nt - 1)
    kbase = ak + (kstart * ldk + head * D) * 2
    vbase = av + (kstart * ldk + head * D) * 2
    for wv in m.waves:
        s = wv.s
        def put64(i, val):
            s[i], s[i + 1] = val & 0xFFFFFFFF, val >> 32
        put64(G.S_Q, aq + head * D * 2)
        put64(G.S_O, ao + head * D * 2)
        put64(G.S_K, kbase)
        put64(G.S_V, vbase)
        s[G.S_LDQ], s[G.S_LDO], s[G.S_LDK] = ldq * 2, ldo * 2, ldk * 2
        s[G.S_ROWS], s[G.S_NT], s[G.S_LASTV] = rows_valid, nt, lastv
        s[G.S_C] = int(E.f2u(c))
        s[G.S_NREC] = (nkeys - 1) * ldk * 2 + D * 2
        if qnorm:
            put64(G.S_SSQ, assq)
            put64(G.S_NW, anw + head * D * 2)
            s[G.S_SSQ_STRIDE], s[G.S_NPART] = nq * 4, nheads
            s[G.S_INVC], s[G.S_EPS] = int(E.f2u(np.float32(1.0 / (nheads * D)))), int(E.f2u(np.float32(eps)))
        wv.v[G.V_TID] = 64 * wv.id + np.arange(64, dtype=np.uint32)
        wv.v[1:] = 0x7FC0BEEF                           # uninitialised registers are NaN poison
        wv.a[:] = 0x7FC0BEEF
    steps = m.run()
    out = bf16_val(mem.get(ao).view(np.uint16).reshape(nq, nheads, D)[:, head])
    qf, kf, vf = bf16_val(qb_[:, head]).astype(np.float64), bf16_val(kb_[kstart:, head]).astype(np.float64), bf16_val(vb_[kstart:, head]).astype(np.float64)
    if qnorm:
        qf = q_eff[:, head].astype(np.float64)
    sc = (qf @ kf.T) * scale
    p = np.exp(sc - sc.max(axis=1, keepdims=True))
    ref = (p / p.sum(axis=1, keepdims=True)) @ vf
    other = mem.get(ao).view(np.uint16).reshape(nq, nheads, D)[:, 1 - head]
    assert (other == 0x7FC0).all(), "the kernel wrote outside its head's columns"
    return out.astype(np.float64), ref, steps, m


def test_generated_text_is_lint_clean_and_assembles(kernel_text, tmp_path):
    assert G.lint(kernel_text) == []
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    src = tmp_path / "k.s"
    src.write_text('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n.text\nkernel:\n' + kernel_text)
    r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "k.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]


@pytest.mark.parametrize("mode", ["lazy", "eager"])
def test_full_block_ragged_keys(kernel_text, mode):
    """256 valid rows, 7 key tiles the last of which holds 20 keys; the cache ends at the last key (staging past it must not
    fault and must not leak into the result)."""
    out, ref, steps, m = run_case(kernel_text, mode, rows_valid=256, nkeys=6 * 64 + 20)
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < 1.2e-2, err


def test_padded_rows_and_idle_waves(kernel_text):
    """72 valid rows (the last q-tile of Lq = 4680): wave 1 has 8 valid rows, waves 2 and 3 only stage and synchronise."""
    out, ref, steps, m = run_case(kernel_text, "lazy", rows_valid=72, nkeys=5 * 64, seed=3)
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < 1.2e-2, err


def test_rescale_path_is_taken_and_exact(kernel_text):
    """Spiked keys far above the running reference (late, early and mid-range; both q-blocks of a wave): the lazy-max slow path
    (O, l, the -m tile and the pending score tile rescaled once, after the pending P.V) must give the fp64 answer."""
    spikes = [(5, 6 * 64 + 10, 3.0), (9, 3, 3.0), (40, 200, 2.5), (200, 130, 3.0)]
    out, ref, steps, m = run_case(kernel_text, "lazy", rows_valid=256, nkeys=7 * 64, seed=5, spikes=spikes)
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < 2e-2, err


@pytest.mark.parametrize("nkeys,mode", [(65, "lazy"), (100, "mixed"), (128, "eager"), (3 * 64 + 1, "mixed")])
def test_two_tiles_is_the_shortest_range(kernel_text, nkeys, mode):
    """The launcher never sends fewer than two key tiles (attention.hip: attn_asm_eligible): the first tile and the last tile are
    both special cases of the generated text and a single tile would have to be both.  Two tiles, ragged or full, must work; the
    third completion model (LDS-DMA lands at once, LDS reads late) is exercised here as well."""
    out, ref, steps, m = run_case(kernel_text, mode, rows_valid=200, nkeys=nkeys, seed=2, kstart=5)
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < 1.2e-2, err


def test_random_geometries(kernel_text):
    """Seeded sweep: row counts around the wave / q-block edges, 2-12 key tiles with every kind of last tile, key ranges that start
    anywhere in the cache, spiked keys that force the rescale path, all three completion models."""
    rng = np.random.default_rng(20261004)
    for it in range(8):
        rows = int(rng.choice([1, 8, 33, 64, 65, 72, 129, 200, 255, 256]))
        nt = int(rng.integers(2, 13))
        nkeys = 64 * (nt - 1) + int(rng.choice([1, 31, 32, 33, 64, int(rng.integers(1, 65))]))
        spikes = [(int(rng.integers(0, rows)), int(rng.integers(0, nkeys)), float(rng.choice([2.5, 4.0, 6.0])))
                  for _ in range(int(rng.integers(0, 4)))]
        mode = str(rng.choice(["lazy", "eager", "mixed"]))
        out, ref, steps, m = run_case(kernel_text, mode, rows_valid=rows, nkeys=nkeys, seed=300 + it, spikes=spikes,
                                      kstart=int(rng.integers(0, 70)))
        err = np.abs(out - ref).max()
        assert np.isfinite(out).all() and err < 2.5e-2, (rows, nkeys, mode, spikes, err)


# ---- QNORM form: RMSNorm of q in the prologue (cross-attention: model.py:172) ---------------------------------------------------
@pytest.fixture(scope="module")
def qn_text():
    return G.generate("buffer", "LLBN", False, True)


def test_qnorm_text_is_lint_clean_and_assembles(qn_text, tmp_path):
    assert G.lint(qn_text) == []
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not os.path.exists(clang):
        pytest.skip("no ROCm assembler here")
    src = tmp_path / "k.s"
    src.write_text('.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n.text\nkernel:\n' + qn_text)
    r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(tmp_path / "k.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:2000]


@pytest.mark.parametrize("rows,nkeys,nheads,head,mode", [(256, 8 * 64, 2, 1, "lazy"), (72, 5 * 64 + 9, 3, 0, "eager"), (200, 2 * 64, 12, 7, "mixed")])
def test_qnorm_prologue(qn_text, rows, nkeys, nheads, head, mode):
    """q normalised in the kernel (sum of the planes in plane order, rsq(sum / C + eps), bf16(bf16(x rinv) w), then the pre-scale)
    against fp64 attention over the separately normalised q: the same bound as the plain form.  12 planes = the 1.3B model's
    N / 128; rows past the valid ones and planes past S_NPART are never read (the buffers are sized exactly)."""
    out, ref, steps, m = run_case(qn_text, mode, rows_valid=rows, nkeys=nkeys, seed=11, nheads=nheads, head=head, qnorm=True)
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < 1.2e-2, err


def test_qnorm_rescale_path(qn_text):
    spikes = [(5, 6 * 64 + 10, 3.0), (40, 200, 2.5), (200, 130, 3.0)]
    out, ref, steps, m = run_case(qn_text, "lazy", rows_valid=256, nkeys=7 * 64, seed=5, spikes=spikes, qnorm=True)
    err = np.abs(out - ref).max()
    assert np.isfinite(out).all() and err < 2e-2, err
