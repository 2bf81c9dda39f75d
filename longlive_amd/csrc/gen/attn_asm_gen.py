#!/usr/bin/env python3
"""Generator of the hand-scheduled self-attention kernel for gfx950 (flash_attn_asm_kernel, longlive_amd/csrc/attention_asm.hip).

Replaces (for long, contiguous key ranges) what flash_attn_pipe_kernel<8, 1> computes: attention() + the sink/window gather of
wan/modules/attention.py:43-197 and causal_model.py:331-360 -- non-causal softmax(scale * Q K^T) V, head_dim 128, bf16 in / out.

Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD" attention; built here without its example file):
  * workgroup = 4 waves = 256 query rows of one head; ONE wave per SIMD, each wave 64 rows (two 32-row q-blocks qb) and the whole
    512-register file, owned by this text (the kernel body is one asm statement; hipcc only loads the arguments):
        a[0:127]    O^T accumulators  o[qb][db]  (db = 32-wide slice of d)            a[64 qb + 16 db + r]
        a[128:191]  Q fragments (B operand of S^T = K Q^T), PRE-SCALED by c = scale*log2(e)  a[128 + 32 qb + 4 ks + j]
        a[192:255]  K fragments of the current tile (A operand), refilled from LDS by ds_read_b128 straight into AGPRs
        v[0:127]    two S^T score buffers  S[x][qb][kb]  (x = tile & 1), 16 registers per 32x32 tile; the packed P^T of a tile is
                    written IN PLACE over registers 0..7 of each of its S tiles (B operand of O^T += V^T P^T)
        v[128:159]  NM[qb][0:15] = -m_ref of the lane's query in all 16 registers: the C operand of the first MFMA of every S chain,
                    so the scores come out of the matrix pipe already relative to the running reference max: p = exp2(S), no VALU
                    subtraction (lazy max, T13: m_ref moves only when a tile's max exceeds it by more than THR)
        v[160:191]  ring of 8 V^T fragments (two ds_read_b64_tr_b16 each)
  * MFMA v_mfma_f32_32x32x16_bf16 in the "swapped" orientation of the shipped kernel (a lane owns one query: softmax is per lane).
  * per 64-key tile i two phases of 32 MFMAs:
        A_i : S(i) = K(i) Q^T            ||  VALU: finish(i-1) = exp2 / row sum / pack of S(i-1)   ||  LDS: first V(i-1) fragments
        B_i : O^T += V(i-1)^T P(i-1)^T   ||  VALU: start(i) = row max of S(i), rescale decision    ||  LDS: K(i+1) -> AGPRs, V(i-1)
                                         ||  LDS-DMA: K(i+3), V(i+2) (buffer_load ... lds, four 1-KiB pieces each per wave)
    one barrier per tile; K and V live in 4-slot LDS rings (128 KiB); waits are counted (vmcnt(8): this tile's pieces stay in flight).
  * fillers are placed by position (gap index) between the MFMAs; LDS waits (lgkmcnt) are inserted by a dependency pass over the
    final instruction stream; a linter checks the software-visible hazards it knows (MFMA result -> VALU, VALU -> MFMA operand, ...).

The same text is executed by tools/gfx950_emu.py in the CPU test-suite (tests/test_attn_asm_emu.py) before it ever reaches a GPU.
"""
from __future__ import annotations

import re
import sys

# ---- fixed register map ---------------------------------------------------------------------------------------------
# inputs (pinned by the HIP wrapper)
S_Q, S_O, S_K, S_V = 8, 10, 12, 14          # 64-bit bases (bytes): Q/O of the workgroup's first row at the head's column; K/V head base
S_LDQ, S_LDO, S_LDK = 16, 17, 18            # row strides in bytes
S_ROWS, S_NT, S_LASTV, S_C, S_NREC = 19, 20, 21, 22, 23
# working scalars
S_KRS, S_VRS = 24, 28                        # buffer descriptors s[24:27], s[28:31]
S_WAVE, S_I, S_STEP, S_T0, S_T1, S_T2, S_T3 = 32, 33, 34, 36, 37, 38, 39
S_M0, S_M1 = 40, 42                          # 64-bit lane masks
S_THR, S_NINF, S_RET, S_KM0, S_VM0, S_MASKI = 48, 49, 50, 52, 53, 54
THR = 8.0                                    # lazy-max threshold in log2 units (P <= 2^8 between rescales)
DMA = "buffer"                               # buffer_load ... lds through a descriptor whose num_records shrinks tile by tile: rows past the key
                                             # range read as zeros, tiles past the end are all zeros (a global_load_lds form with clamped tiles /
                                             # rows existed and staged identical bytes 1.5 % slower: round-3 notes in DESIGN.md)
PFX = "LL"                                   # label prefix (one per kernel in the translation unit)
DIAG = False                                 # diagnostic build only (tools/attn_asm_diag.hip): s_memtime stamps around the phases of every
                                             # tile, summed per wave and stored to a debug buffer (s[56:57] + 32 * (4 * s58 + wave)); never in the library
S_DBG, S_WG = 56, 58
# QNORM form (flash_attn_asm_qn_kernel): Q arrives as the RAW output of the q projection and is RMS-normalised in the prologue
# (WanRMSNorm over all C = H * 128 channels of the row, wan/modules/model.py:78-86, as cross-attention applies it to q: :172) from the
# per-(n-tile, row) sums of squares the projection's epilogue left (gemm_asm_128_bias_ssq):
S_SSQ, S_SSQ_STRIDE, S_NPART, S_NW, S_INVC, S_EPS = 80, 82, 83, 84, 86, 87      # s[80:81] sums at the workgroup's first row (fp32, bytes); bytes between
                                             # n-tile planes; planes; s[84:85] norm weight at the head's first channel (bf16); 1 / C; eps (fp32 bits)
S_C2 = 44                                    # (c, c) in s[44:45]: packed-multiply operand
NPART_MAX = 16                               # planes summed (absent ones read as +0: x + 0 is exact), in plane order
import os as _os

# ---- schedule / timing knobs (environment ASM_<name>, ASM_G_<name> for the GEMM generator) ---------------------------------
# Two kinds.  Schedule knobs (VF_LEAD, FIN_SPAN, ...) move instructions and keep the results valid; the ones in effect are recorded
# in KNOBS_SET and travel into the library's ll_*_plan strings (knobs_header).  TIMING-ONLY knobs (NO_*: a part of the loop is
# removed, the results are INVALID) are refused unless the generator was started with --diag (tools/build_diag_variants.sh,
# tools/build_gemm_variant.sh) or a tool called allow_diag_knobs(): a variable left over in a shell can never build a wrong kernel
# into the shipped library.  The Makefile runs the generators in an EMPTY environment on top of that.
DIAG_KNOBS_OK = "--diag" in sys.argv
KNOBS_SET = {}


def allow_diag_knobs(ok=True):
    global DIAG_KNOBS_OK
    DIAG_KNOBS_OK = ok


def knob_env(prefix, name, default):
    raw = _os.environ.get(prefix + name)
    if raw is None:
        return default
    val = type(default)(raw)
    if val == default:
        return default
    if name.startswith("NO_") and not DIAG_KNOBS_OK:
        raise SystemExit(f"{prefix}{name}={raw}: timing-only knob (results invalid) refused without --diag; "
                         f"unset it, or build a diagnostic variant with tools/build_diag_variants.sh / tools/build_gemm_variant.sh")
    KNOBS_SET[prefix + name] = raw
    return val


def KNOB(name, default):
    return knob_env("ASM_", name, default)


def knobs_string():
    """'' for the default schedule, else 'NAME=value,...#crc' of every ASM_* / ASM_G_* variable of this environment that a generator
    would honour (the names are taken from the environment: both generators read theirs lazily)."""
    import zlib
    items = sorted((k, v) for k, v in _os.environ.items() if k.startswith("ASM_") and v not in ("", "0"))
    if not items:
        return ""
    txt = ",".join(f"{k}={v}" for k, v in items)
    return f"{txt}#{zlib.crc32(txt.encode()) & 0xffffffff:08x}"


def knobs_header(path):
    ks = knobs_string()
    bad = [k for k in ks.split("#")[0].split(",") if "_NO_" in k.split("=")[0]] if ks else []
    if bad and not DIAG_KNOBS_OK:
        raise SystemExit(f"{bad}: timing-only knobs refused without --diag")
    open(path, "w").write("// generated by gen/attn_asm_gen.py --knobs-header: the generator knobs this library was built with\n"
                          f"#define LL_ASM_KNOBS \"{ks}\"\n")


LSUM = KNOB("LSUM", 0)                       # 1: row sums by the matrix pipe (ones-MFMA into LACC); 0: by v_add_f32 into 4 partials per q-block
DOT_LSUM = KNOB("DOT_LSUM", 0)               # LSUM = 0: row sums by v_dot2c_f32_bf16 (l += P.lo * 1 + P.hi * 1) on the PACKED bf16 P registers: 32 per tile
                                             # instead of 64 v_add_f32, four accumulators per q-block as before; sums what P.V multiplies
S_ONES = 46                                  # (1.0, 1.0) in bf16 for DOT_LSUM
PK_LSUM = KNOB("PK_LSUM", 0)                 # LSUM = 0: row sums by v_pk_add_f32 on element pairs (two pair-accumulators per q-block) instead of v_add_f32 per element
ADD_LATE = KNOB("ADD_LATE", 0)               # LSUM = 0: the adds of the elements whose registers survive the in-place pack (registers
                                             # 8..15 of every score tile) are issued in phase B instead of phase A
DIAG_PRO = KNOB("DIAG_PRO", 0)               # with --diag: instead of the per-phase sums, eight s_memrealtime stamps (100 MHz) of the kernel's TIMELINE per wave
                                             # (entry, staging issued, Q + tiles landed, Q converted, loop start, loop end, stores issued, stores done) -> dbg[64 B]
S_TS, S_ACC = 60, 64                         # last stamp s[60:61]; sums s[64:65] A, s[66:67] B, s[68:69] wait+barrier, s[70:71] whole loop

A_O, A_Q, A_K = 0, 128, 192
NVF = KNOB("NVF", 4 if LSUM else 8)          # ring of V^T fragments (4 registers each); the row-sum MFMA form needs the room for LACC
V_S, V_NM, V_VF = 0, 128, 160                # S buffers 0..127, -m tiles 128..159, V^T fragment ring from 160
_next = V_VF + 4 * NVF
V_L = V_LACC = _next                          # LSUM = 0: l[qb][4] partial row sums (8 registers); LSUM = 1: LACC[qb][0:15], D of the row-sum
_next += 32 if LSUM else 8                    # MFMA (every register of a lane = its query's sum)
V_KOFF, V_VOFF, V_DK, V_DV = _next, _next + 8, _next + 12, _next + 16      # 8 K-read offsets, 4 V-read offsets, 4 + 4 DMA source offsets
V_ONES = _next + 20                           # 4 registers of bf16 1.0 (A operand of the row-sum MFMA)
V_MX = _next + 24                             # mx[qb]: the tile's row max (both halves)
V_ROW = _next + 26                            # per-qb row index within the workgroup
V_LANE, V_R, V_H = _next + 28, _next + 29, _next + 30
V_T = _next + 32                              # 14 temporaries
assert V_T + 14 <= 256, V_T
V_TID = 0                                     # input: workitem id

KSLOT = lambda s: 16384 * (s & 3)             # LDS: K ring at 0, V ring at 64 KiB
VSLOT = lambda s: 65536 + 16384 * (s & 3)


def sreg(i, n=1):
    return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


def vreg(i, n=1):
    return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"


def areg(i, n=1):
    return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"


def S_t(x, qb, kb):
    return V_S + 64 * x + 32 * qb + 16 * kb


def P_f(x, qb, kstep):                       # packed P^T fragment (4 registers) of k-step (kb, s2) = (kstep >> 1, kstep & 1)
    return S_t(x, qb, kstep >> 1) + 4 * (kstep & 1)


class Gen:
    def __init__(self):
        self.out = []
        self.uid = 0

    def I(self, text):
        self.out.append(text)

    def L(self, name):
        self.out.append(name + ":")

    def label(self, stem):
        self.uid += 1
        return f"LL_{stem}_{self.uid}"

    # ---- phase interleaver ------------------------------------------------------------------------------------------
    def phase(self, mfmas, fillers):
        """mfmas: list of instruction texts; fillers: list of (position, text) -- a filler at position p is emitted after MFMA
        floor(p) (p < 0: before the first one); equal positions keep their order."""
        fl = sorted(enumerate(fillers), key=lambda t: (t[1][0], t[0]))
        k = 0
        while k < len(fl) and fl[k][1][0] < 0:
            self.I(fl[k][1][1]); k += 1
        for g, m in enumerate(mfmas):
            self.I(m)
            while k < len(fl) and fl[k][1][0] < g + 1:
                self.I(fl[k][1][1]); k += 1
        while k < len(fl):
            self.I(fl[k][1][1]); k += 1


# ---- building blocks ------------------------------------------------------------------------------------------------
def qk_mfmas(x):
    """S(x) = K Q^T: ks outer, then kb, qb -- every K fragment feeds two consecutive MFMAs; first k-step starts from NM = -m_ref."""
    out = []
    for ks in range(8):
        for kb in range(2):
            for qb in range(2):
                d = vreg(S_t(x, qb, kb), 16)
                c = vreg(V_NM + 16 * qb, 16) if ks == 0 else d
                out.append(f"v_mfma_f32_32x32x16_bf16 {d}, {areg(A_K + 32 * kb + 4 * ks, 4)}, {areg(A_Q + 32 * qb + 4 * ks, 4)}, {c}")
    return out


def pv_mfmas(y):
    """O^T += V^T P^T with P of buffer y: k-step outer, db, qb -- every V fragment (ring slot n % NVF) feeds two consecutive MFMAs.
    After the eight MFMAs of a k-step, two more sum its P over the keys on the matrix pipe: A = all ones, so every row of the
    32 x 32 result -- every register of a lane -- is the lane's query's sum over the 16 keys of BOTH halves (no VALU adds, no
    cross-half exchange, and the sum is taken of the bf16 values the P.V product uses)."""
    out = []
    for kstep in range(4):
        for db in range(4):
            n = 4 * kstep + db
            for qb in range(2):
                o = areg(A_O + 64 * qb + 16 * db, 16)
                out.append(f"v_mfma_f32_32x32x16_bf16 {o}, {vreg(V_VF + 4 * (n % NVF), 4)}, {vreg(P_f(y, qb, kstep), 4)}, {o}")
        for qb in range(2 if LSUM else 0):
            l = vreg(V_LACC + 16 * qb, 16)
            out.append(f"v_mfma_f32_32x32x16_bf16 {l}, {vreg(V_ONES, 4)}, {vreg(P_f(y, qb, kstep), 4)}, {l}")
    return out


def pv_index(n):
    """index of the first of the two MFMAs that consume V^T fragment n"""
    return (10 if LSUM else 8) * (n >> 2) + 2 * (n & 3)


def v_frag_reads(n, vslot):
    """the two transposed reads of V^T fragment n = (kstep, db) of the tile in ring slot vslot"""
    kstep, db = n >> 2, n & 3
    base = VSLOT(vslot) - 65536 + (32 * (kstep >> 1) + 16 * (kstep & 1)) * 256      # V_VOFF registers carry the +65536
    d = V_VF + 4 * (n % NVF)
    return [f"ds_read_b64_tr_b16 {vreg(d, 2)}, {vreg(V_VOFF + db)} offset:{base}",
            f"ds_read_b64_tr_b16 {vreg(d + 2, 2)}, {vreg(V_VOFF + db)} offset:{base + 8 * 256}"]


def k_frag_reads(kslot):
    out = []
    for ks in range(8):
        for kb in range(2):
            out.append(f"ds_read_b128 {areg(A_K + 32 * kb + 4 * ks, 4)}, {vreg(V_KOFF + ks)} offset:{KSLOT(kslot) + 8192 * kb}")
    return out


def finish_ops(y, with_pos=False):
    """exp2 and pack of S(y) -> P in place (the row sum is taken by the matrix pipe: pv_mfmas).  Order = the order PV consumes the fragments: k-step, then qb.  Emitted as a
    software pipeline (exp of element n, add of element n - DA, pack of a pair DC back) so that no instruction depends on its
    near predecessors.  with_pos: (element index the op belongs to, text) so that the caller can place ops by deadline."""
    DC = KNOB("FIN_DC", 5 if not LSUM else 3)
    elems = []
    for kstep in range(4):
        for qb in range(2):
            base = S_t(y, qb, kstep >> 1) + 8 * (kstep & 1)
            for j in range(8):
                elems.append((qb, base + j, P_f(y, qb, kstep) + (j >> 1), j))
    ops = []
    n = len(elems)
    noexp = KNOB("NO_EXP", 0)
    for t in range(n + DC + 2):
        if t < n:
            ops.append((t, (f"v_mov_b32 {vreg(elems[t][1])}, {vreg(elems[t][1])}" if noexp else f"v_exp_f32 {vreg(elems[t][1])}, {vreg(elems[t][1])}")))
        if not LSUM and DOT_LSUM:
            pass                                          # (the sums ride on the packed registers: below, one slot after each cvt_pk)
        elif not LSUM and 2 <= t < n + 2:
            qb, r, _, j = elems[t - 2]
            late = ADD_LATE and (r & 15) >= 8
            if PK_LSUM:                                   # one packed add per PAIR of elements (after the second one's exp): 32 instead of 64 per tile
                if j & 1:
                    a = V_L + 4 * qb + 2 * ((j >> 1) & 1)
                    ops.append((t if not late else 1000 + t, f"v_pk_add_f32 {vreg(a, 2)}, {vreg(a, 2)}, {vreg(r - 1, 2)}"))
            else:
                ops.append((t if not late else 1000 + t, f"v_add_f32 {vreg(V_L + 4 * qb + (j & 3))}, {vreg(V_L + 4 * qb + (j & 3))}, {vreg(r)}"))
        if t >= DC and (t - DC) % 2 == 0 and t - DC < n:
            qb, r0, dst, j = elems[t - DC]
            ops.append((t, f"v_cvt_pk_bf16_f32 {vreg(dst)}, {vreg(r0)}, {vreg(r0 + 1)}"))
        if DOT_LSUM and not LSUM and t >= DC + 1 and (t - DC - 1) % 2 == 0 and t - DC - 1 < n:
            qb, r0, dst, j = elems[t - DC - 1]
            ops.append((t, f"v_dot2c_f32_bf16 {vreg(V_L + 4 * qb + ((j >> 1) & 3))}, {sreg(S_ONES)}, {vreg(dst)}"))
    if KNOB("NO_FIN", 0):
        ops = []
    return ops if with_pos else [o for _, o in ops]


def start_ops(x):
    """row max of S(x) per q-block (two chains per q-block, merged), cross-half exchange, rescale decision masks in S_M0 / S_M1"""
    ops = []
    chains = []
    finals = []
    for qb in range(2):
        for kb in range(2):
            b = S_t(x, qb, kb)                        # 16-aligned: register b + j sits in VGPR bank j & 3
            ci = 2 * qb + kb
            ta, tb = V_T + 6 + 2 * ci, V_T + 7 + 2 * ci
            # (ta in VGPR bank 1 or 2, read next to operands in banks 3, 0; tb in bank 0 or 3, next to banks 1, 2 -- measured: no effect)
            cur = ta
            c = [f"v_max3_f32 {vreg(cur)}, {vreg(b)}, {vreg(b + 1)}, {vreg(b + 2)}"]
            for j in range(3, 15, 2):
                nxt = tb if cur == ta else ta
                c.append(f"v_max3_f32 {vreg(nxt)}, {vreg(cur)}, {vreg(b + j)}, {vreg(b + j + 1)}")
                cur = nxt
            nxt = tb if cur == ta else ta
            c.append(f"v_max_f32 {vreg(nxt)}, {vreg(cur)}, {vreg(b + 15)}")
            finals.append(nxt)
            chains.append(c)
    for step in range(len(chains[0])):               # round-robin over the four chains: 3 independent ops between dependents
        for c in chains:
            ops.append(c[step])
    for qb in range(2):
        ops.append(f"v_max_f32 {vreg(V_MX + qb)}, {vreg(finals[2 * qb])}, {vreg(finals[2 * qb + 1])}")
    for qb in range(2):
        ops.append(f"v_mov_b32 {vreg(V_T + 4 + qb)}, {vreg(V_MX + qb)}")
    ops.append("s_nop 0")
    for qb in range(2):                               # lanes 32-63 of MX <-> lanes 0-31 of the copy: each then holds both halves' maxima
        ops.append(f"v_permlane32_swap_b32 {vreg(V_MX + qb)}, {vreg(V_T + 4 + qb)}")
    for qb in range(2):
        ops.append(f"v_max_f32 {vreg(V_MX + qb)}, {vreg(V_MX + qb)}, {vreg(V_T + 4 + qb)}")
    ops.append(f"v_cmp_gt_f32_e64 {sreg(S_M0, 2)}, {vreg(V_MX)}, {sreg(S_THR)}")
    ops.append(f"v_cmp_gt_f32_e64 {sreg(S_M1, 2)}, {vreg(V_MX + 1)}, {sreg(S_THR)}")
    return ops


def dma_pieces(which, slot):
    """this wave's four 1-KiB LDS-DMA pieces of one K or V tile into ring slot `slot`.  M0_OFFS = 1: ONE m0 write per tensor, the
    pieces addressed by the instruction's 12-bit offset, which moves the LDS destination AND the source address by 1024 i -- the
    source offsets V_DK / V_DV are pre-biased by -1024 i (the range check sees voffset + offset: unchanged)."""
    ops = []
    m0b, ring, off, rs = (S_KM0, KSLOT, V_DK, S_KRS) if which == "K" else (S_VM0, VSLOT, V_DV, S_VRS)
    if KNOB("M0_OFFS", 0):
        ops.append(f"s_add_u32 m0, {sreg(m0b)}, {ring(slot)}")
        for i in range(4):
            ops.append(f"buffer_load_dwordx4 {vreg(off + i)}, {sreg(rs, 4)}, 0 offen" + (f" offset:{1024 * i}" if i else "") + " lds")
        return ops
    for i in range(4):
        ops.append(f"s_add_u32 m0, {sreg(m0b)}, {ring(slot) + 1024 * i}")
        ops.append(f"buffer_load_dwordx4 {vreg(off + i)}, {sreg(rs, 4)}, 0 offen lds")
    return ops


def dma_step(which):
    """advance the tile source to the next tile: base += one tile, num_records -= one tile (floored at 0: past the end every
    lane is out of range and the hardware returns zeros)"""
    rs = S_KRS if which == "K" else S_VRS
    return [f"s_add_u32 {sreg(rs)}, {sreg(rs)}, {sreg(S_STEP)}", f"s_addc_u32 {sreg(rs + 1)}, {sreg(rs + 1)}, 0",
            f"s_sub_i32 {sreg(rs + 2)}, {sreg(rs + 2)}, {sreg(S_STEP)}", f"s_max_i32 {sreg(rs + 2)}, {sreg(rs + 2)}, 0"]


def dma_ops(kslot, vslot):
    """LDS-DMA of one K tile and one V tile + the step to the next tile"""
    return dma_pieces("K", kslot) + dma_pieces("V", vslot) + dma_step("K") + dma_step("V")


def spread(ops, lo, hi):
    n = len(ops)
    return [(lo + (hi - lo) * k / max(1, n), op) for k, op in enumerate(ops)]


# ---- slow paths (not interleaved; rare) -------------------------------------------------------------------------------
def gen_rescale(g: Gen, x: int, ret_labels):
    """Some query's tile max exceeded m_ref + THR (S_M0 | S_M1 != 0).  Runs AFTER every P.V of the pending tile has issued
    (T13's safe order).  Per lane: d = max(mx, 0); f = 2^-d; O *= f; l *= f; NM -= d (m_ref += d); S(x) -= d."""
    g.I("s_nop 15")
    g.I("s_nop 15")                                   # the last P.V / row-sum MFMAs have written their accumulators
    for qb in range(2):
        d, f = V_T + qb, V_T + 2 + qb
        g.I(f"v_max_f32 {vreg(d)}, {vreg(V_MX + qb)}, 0")
        g.I(f"v_exp_f32_e64 {vreg(f)}, -{vreg(d)}")
    for qb in range(2):
        d, f = V_T + qb, V_T + 2 + qb
        if LSUM:
            g.I(f"v_mul_f32 {vreg(V_LACC + 16 * qb)}, {vreg(V_LACC + 16 * qb)}, {vreg(f)}")      # only register 0 is read at the end
        else:
            for k in range(4):
                g.I(f"v_mul_f32 {vreg(V_L + 4 * qb + k)}, {vreg(V_L + 4 * qb + k)}, {vreg(f)}")
        for r in range(16):
            g.I(f"v_sub_f32 {vreg(V_NM + 16 * qb + r)}, {vreg(V_NM + 16 * qb + r)}, {vreg(d)}")
        for kb in range(2):
            for r in range(16):
                g.I(f"v_sub_f32 {vreg(S_t(x, qb, kb) + r)}, {vreg(S_t(x, qb, kb) + r)}, {vreg(d)}")
        for r in range(0, 64, 4):                    # O through four temporaries at a time
            for k in range(4):
                g.I(f"v_accvgpr_read_b32 {vreg(V_T + 10 + k)}, {areg(A_O + 64 * qb + r + k)}")
            for k in range(4):
                g.I(f"v_mul_f32 {vreg(V_T + 10 + k)}, {vreg(V_T + 10 + k)}, {vreg(f)}")
            for k in range(4):
                g.I(f"v_accvgpr_write_b32 {areg(A_O + 64 * qb + r + k)}, {vreg(V_T + 10 + k)}")
    g.I("s_nop 7")                                    # accvgpr / VALU writes -> the next MFMA's operands
    ret_dispatch(g, ret_labels)


def gen_mask(g: Gen, x: int, ret_labels):
    """Ragged last tile: keys >= last_valid get score -inf.  A lane (r, h) holds key 32 kb + (j & 3) + 8 (j >> 2) + 4 h in register j."""
    g.I("s_nop 15")
    g.I("s_nop 15")                                   # S(x) has left the matrix pipe
    g.I(f"v_lshlrev_b32 {vreg(V_T + 6)}, 2, {vreg(V_H)}")
    g.I(f"v_sub_u32 {vreg(V_T + 6)}, {sreg(S_LASTV)}, {vreg(V_T + 6)}")          # last_valid - 4 h
    g.I(f"v_mov_b32 {vreg(V_T + 7)}, {sreg(S_NINF)}")                            # (one SGPR source per VALU instruction)
    for kb in range(2):
        for j in range(16):
            key = 32 * kb + (j & 3) + 8 * (j >> 2)
            g.I(f"v_cmp_gt_i32_e64 {sreg(S_M0, 2)}, {vreg(V_T + 6)}, {key}")     # valid <=> last_valid - 4 h > key0
            for qb in range(2):
                r = S_t(x, qb, kb) + j
                g.I(f"v_cndmask_b32_e64 {vreg(r)}, {vreg(V_T + 7)}, {vreg(r)}, {sreg(S_M0, 2)}")
    ret_dispatch(g, ret_labels)


def ret_dispatch(g: Gen, ret_labels):
    for k, lab in enumerate(ret_labels[:-1]):
        g.I(f"s_cmp_eq_u32 {sreg(S_RET)}, {k}")
        g.I(f"s_cbranch_scc1 {lab}")
    g.I(f"s_branch {ret_labels[-1]}")


# ---- the kernel -------------------------------------------------------------------------------------------------------
def generate(dma: str = "buffer", prefix: str = "LL", diag: bool = False, qnorm: bool = False) -> str:
    global DMA, PFX, DIAG
    assert dma == "buffer"
    DMA, PFX, DIAG = dma, prefix, diag
    g = Gen()
    I = g.I
    # ================= setup =================
    pstamp(g, 0)
    I(f"v_and_b32 {vreg(V_LANE)}, 63, {vreg(V_TID)}")
    I(f"v_lshrrev_b32 {vreg(V_T)}, 6, {vreg(V_TID)}")
    I("s_nop 0")
    I(f"v_readfirstlane_b32 {sreg(S_WAVE)}, {vreg(V_T)}")
    I(f"v_and_b32 {vreg(V_R)}, 31, {vreg(V_LANE)}")
    I(f"v_lshrrev_b32 {vreg(V_H)}, 5, {vreg(V_LANE)}")
    I(f"s_mov_b32 {sreg(S_THR)}, {hex(f32bits(THR))}")
    if DOT_LSUM:
        I(f"s_mov_b32 {sreg(S_ONES)}, 0x3f803f80")
    I(f"s_mov_b32 {sreg(S_NINF)}, 0xff800000")
    I(f"s_lshl_b32 {sreg(S_STEP)}, {sreg(S_LDK)}, 6")                       # bytes per 64-key tile
    I(f"s_sub_u32 {sreg(S_MASKI)}, {sreg(S_NT)}, 1")
    I(f"s_cmp_lt_u32 {sreg(S_LASTV)}, 64")
    I(f"s_cselect_b32 {sreg(S_MASKI)}, {sreg(S_MASKI)}, -1")
    # buffer descriptors: base, stride 0, num_records, raw dword format (0x00020000 = the compiler's make_buffer_rsrc flags)
    for rs, base in ((S_KRS, S_K), (S_VRS, S_V)):
        I(f"s_mov_b32 {sreg(rs)}, {sreg(base)}")
        I(f"s_and_b32 {sreg(rs + 1)}, {sreg(base + 1)}, 0xffff")
        I(f"s_mov_b32 {sreg(rs + 2)}, {sreg(S_NREC)}")
        I(f"s_mov_b32 {sreg(rs + 3)}, 0x00020000")
    # LDS-DMA destination of this wave's pieces: slot + (4 wave + i) KiB
    I(f"s_lshl_b32 {sreg(S_KM0)}, {sreg(S_WAVE)}, 12")
    I(f"s_mov_b32 {sreg(S_VM0)}, {sreg(S_KM0)}")
    # DMA source offsets: piece i covers keys 16 w + 4 i + (lane >> 4); the lane at LDS position pos = lane & 15 fetches chunk
    # pos ^ (key & 15) of K and pos ^ ((key & 3) << 2) of V (swizzle on the SOURCE side; the LDS image is lane-linear)
    t_g, t_pos, t_key, t_x = V_T, V_T + 1, V_T + 2, V_T + 3
    I(f"v_lshrrev_b32 {vreg(t_g)}, 4, {vreg(V_LANE)}")
    I(f"v_and_b32 {vreg(t_pos)}, 15, {vreg(V_LANE)}")
    I(f"s_lshl_b32 {sreg(S_T0)}, {sreg(S_WAVE)}, 4")
    for i in range(4):
        I(f"v_add_u32 {vreg(t_key)}, {sreg(S_T0)}, {vreg(t_g)}")
        I(f"v_add_u32 {vreg(t_key)}, {4 * i}, {vreg(t_key)}")                                    # key within the tile
        I(f"v_and_b32 {vreg(t_x)}, 15, {vreg(t_key)}")
        I(f"v_xor_b32 {vreg(t_x)}, {vreg(t_x)}, {vreg(t_pos)}")
        I(f"v_lshlrev_b32 {vreg(t_x)}, 4, {vreg(t_x)}")
        I(f"v_mul_lo_u32 {vreg(V_DK + i)}, {vreg(t_key)}, {sreg(S_LDK)}")
        I(f"v_add_u32 {vreg(V_DK + i)}, {vreg(V_DK + i)}, {vreg(t_x)}")
        I(f"v_lshlrev_b32 {vreg(t_x)}, 2, {vreg(t_g)}")
        I(f"v_xor_b32 {vreg(t_x)}, {vreg(t_x)}, {vreg(t_pos)}")
        I(f"v_lshlrev_b32 {vreg(t_x)}, 4, {vreg(t_x)}")
        I(f"v_mul_lo_u32 {vreg(V_DV + i)}, {vreg(t_key)}, {sreg(S_LDK)}")
        I(f"v_add_u32 {vreg(V_DV + i)}, {vreg(V_DV + i)}, {vreg(t_x)}")
        if KNOB("M0_OFFS", 0) and i:
            I(f"v_subrev_u32 {vreg(V_DK + i)}, {1024 * i}, {vreg(V_DK + i)}")
            I(f"v_subrev_u32 {vreg(V_DV + i)}, {1024 * i}, {vreg(V_DV + i)}")
    # prologue staging: K(0..3), V(0..2)
    for t in range(4):
        for op in dma_pieces("K", t) + dma_step("K"):
            I(op)
        if t < 3:
            for op in dma_pieces("V", t) + dma_step("V"):
                I(op)
    pstamp(g, 1)
    # idle waves (all 64 rows of the wave are padding) only stage and synchronise
    I(f"s_lshl_b32 {sreg(S_T0)}, {sreg(S_WAVE)}, 6")
    I(f"s_cmp_ge_u32 {sreg(S_T0)}, {sreg(S_ROWS)}")
    I("s_cbranch_scc1 LL_IDLE")
    # K-read offsets: key r (+32 kb), chunk (2 ks + h) ^ (r & 15)
    I(f"v_and_b32 {vreg(V_T)}, 15, {vreg(V_R)}")
    I(f"v_lshlrev_b32 {vreg(V_T + 1)}, 8, {vreg(V_R)}")
    for ks in range(8):
        I(f"v_add_u32 {vreg(V_T + 2)}, {2 * ks}, {vreg(V_H)}")
        I(f"v_xor_b32 {vreg(V_T + 2)}, {vreg(V_T + 2)}, {vreg(V_T)}")
        I(f"v_lshl_add_u32 {vreg(V_KOFF + ks)}, {vreg(V_T + 2)}, 4, {vreg(V_T + 1)}")
    # V transposed-read offsets: lane = 16 g + 4 tq + tp supplies row 4 h + tq (+ 32 kb + 16 s2 [+ 8]), columns 32 db + 16 (g & 1) + 4 tp
    tq, tp, tg1, trow = V_T, V_T + 1, V_T + 2, V_T + 3
    I(f"v_and_b32 {vreg(tq)}, 15, {vreg(V_LANE)}")
    I(f"v_lshrrev_b32 {vreg(tq)}, 2, {vreg(tq)}")
    I(f"v_and_b32 {vreg(tp)}, 3, {vreg(V_LANE)}")
    I(f"v_lshrrev_b32 {vreg(tg1)}, 4, {vreg(V_LANE)}")
    I(f"v_and_b32 {vreg(tg1)}, 1, {vreg(tg1)}")
    I(f"v_lshl_add_u32 {vreg(trow)}, {vreg(V_H)}, 2, {vreg(tq)}")            # 4 h + tq
    I(f"v_lshlrev_b32 {vreg(trow)}, 8, {vreg(trow)}")                         # * 256
    I(f"v_add_u32 {vreg(trow)}, 0x10000, {vreg(trow)}")                      # the V ring starts at 64 KiB
    I(f"v_lshrrev_b32 {vreg(V_T + 4)}, 1, {vreg(tp)}")                        # tp >> 1
    I(f"v_lshl_add_u32 {vreg(V_T + 4)}, {vreg(tg1)}, 1, {vreg(V_T + 4)}")    # 2 tg1 + (tp >> 1)
    I(f"v_lshlrev_b32 {vreg(V_T + 5)}, 2, {vreg(tq)}")                        # swizzle term tq << 2
    I(f"v_and_b32 {vreg(V_T + 6)}, 1, {vreg(tp)}")
    I(f"v_lshlrev_b32 {vreg(V_T + 6)}, 3, {vreg(V_T + 6)}")                   # (tp & 1) * 8
    for db in range(4):
        I(f"v_add_u32 {vreg(V_T + 7)}, {4 * db}, {vreg(V_T + 4)}")
        I(f"v_xor_b32 {vreg(V_T + 7)}, {vreg(V_T + 7)}, {vreg(V_T + 5)}")
        I(f"v_lshl_add_u32 {vreg(V_T + 7)}, {vreg(V_T + 7)}, 4, {vreg(V_T + 6)}")
        I(f"v_add_u32 {vreg(V_VOFF + db)}, {vreg(V_T + 7)}, {vreg(trow)}")
    # ---- Q: rows wave*64 + 32 qb + r (clamped), 16 bytes at d = 16 ks + 8 h; pre-scale by c; -> AGPRs
    for qb in range(2):
        row = V_ROW + qb
        I(f"v_add_u32 {vreg(row)}, {sreg(S_T0)}, {vreg(V_R)}")
        if qb:
            I(f"v_add_u32 {vreg(row)}, 32, {vreg(row)}")
    I(f"s_sub_u32 {sreg(S_T1)}, {sreg(S_ROWS)}, 1")
    for qb in range(2):
        I(f"v_min_u32 {vreg(V_T + qb)}, {sreg(S_T1)}, {vreg(V_ROW + qb)}")
        if qnorm:
            I(f"v_lshlrev_b32 {vreg(V_T + 2 + qb)}, 2, {vreg(V_T + qb)}")                # byte offset of the (clamped) row's sum inside a plane
        I(f"v_mul_lo_u32 {vreg(V_T + qb)}, {vreg(V_T + qb)}, {sreg(S_LDQ)}")
        I(f"v_lshl_add_u32 {vreg(V_T + qb)}, {vreg(V_H)}, 4, {vreg(V_T + qb)}")      # + 8 h elements = 16 h bytes
        for ks in range(8):
            I(f"global_load_dwordx4 {vreg(64 * 0 + 32 * qb + 4 * ks, 4)}, {vreg(V_T + qb)}, {sreg(S_Q, 2)} offset:{32 * ks}")
    if qnorm:
        gen_qnorm_loads(g)
    # O = 0, l = 0 while the loads fly
    for r in range(128):
        I(f"v_accvgpr_write_b32 {areg(A_O + r)}, 0")
    for k in range(32 if LSUM else 8):
        I(f"v_mov_b32 {vreg(V_LACC + k)}, 0")
    if LSUM:
        for k in range(4):
            I(f"v_mov_b32 {vreg(V_ONES + k)}, 0x3f803f80")
    I("s_waitcnt vmcnt(0)")                           # Q, and every staged tile of the prologue
    pstamp(g, 2)
    if qnorm:
        gen_qnorm_convert(g)
    for qb in range(0 if qnorm else 2):
        for ks in range(8):
            for j in range(4):
                src = 32 * qb + 4 * ks + j
                lo, hi = V_T + 4, V_T + 5
                I(f"v_lshlrev_b32 {vreg(lo)}, 16, {vreg(src)}")
                I(f"v_and_b32 {vreg(hi)}, 0xffff0000, {vreg(src)}")
                I(f"v_mul_f32 {vreg(lo)}, {sreg(S_C)}, {vreg(lo)}")
                I(f"v_mul_f32 {vreg(hi)}, {sreg(S_C)}, {vreg(hi)}")
                I(f"v_cvt_pk_bf16_f32 {vreg(lo)}, {vreg(lo)}, {vreg(hi)}")
                I(f"v_accvgpr_write_b32 {areg(A_Q + 32 * qb + 4 * ks + j)}, {vreg(lo)}")
    pstamp(g, 3)
    I("s_barrier")                                    # (1) every wave's share of K(0..3), V(0..2) is in LDS
    # ---- tile 0: K(0) -> AGPRs, S(0) from zero, exact row max -> m_ref, NM, S -= m_ref
    for op in k_frag_reads(0):
        I(op)
    I("s_waitcnt lgkmcnt(0)")
    I("s_nop 7")
    for k, m in enumerate(qk_mfmas(0)):
        I(m if k >= 4 else re.sub(r", v\[\d+:\d+\]$", ", 0", m))          # first k-step accumulates from zero
    I("s_nop 15")
    I("s_nop 15")
    for op in start_ops(0)[:-2]:                      # maxima only (no threshold test on the first tile)
        I(op)
    for qb in range(2):
        for r in range(16):
            I(f"v_sub_f32 {vreg(V_NM + 16 * qb + r)}, 0, {vreg(V_MX + qb)}")
        for kb in range(2):
            for r in range(16):
                I(f"v_sub_f32 {vreg(S_t(0, qb, kb) + r)}, {vreg(S_t(0, qb, kb) + r)}, {vreg(V_MX + qb)}")
    for op in k_frag_reads(1):                        # K(1) -> AGPRs (K(0)'s MFMAs have all issued)
        I(op)
    I("s_waitcnt lgkmcnt(0)")
    I("s_barrier")                                    # (2) K(0) has been read by every wave: its slot may be refilled
    I(f"s_mov_b32 {sreg(S_I)}, 1")
    pstamp(g, 4)
    if DIAG and not DIAG_PRO:
        for k in range(8):
            I(f"s_mov_b32 {sreg(S_ACC + k)}, 0")
        stamp(g)
        I(f"s_mov_b64 s[72:73], {sreg(S_TS, 2)}")
    # ================= main loop, unrolled by 4 (ring slots and score-buffer parity are static per copy) =================
    g.L("LL_LOOP")
    for u in (1, 2, 3, 0):
        gen_tile(g, u)
        I(f"s_add_u32 {sreg(S_I)}, {sreg(S_I)}, 1")
        I(f"s_cmp_ge_u32 {sreg(S_I)}, {sreg(S_NT)}")
        I(f"s_cbranch_scc1 LL_TAIL_{u}")
    I("s_branch LL_LOOP")
    # out-of-line blocks
    for x in range(2):
        g.L(f"LL_RESCALE_{x}")
        gen_rescale(g, x, [f"LL_RESC_RET_{u}" for u in (x, x + 2)][::1] if False else rescale_rets(x))
        g.L(f"LL_MASK_{x}")
        gen_mask(g, x, mask_rets(x))
    # ================= tails: finish and P.V of the last tile (buffer (nt-1) & 1, V slot (nt-1) & 3) =================
    for u in (1, 2, 3, 0):
        g.L(f"LL_TAIL_{u}")
        y = u & 1                                     # the last tile's index is congruent to u mod 4
        for op in finish_ops(y):
            I(op)
        for n in range(NVF):
            for op in v_frag_reads(n, u):
                I(op)
        fillers = []
        for n in range(NVF, 16):
            for k, op in enumerate(v_frag_reads(n, u)):
                fillers.append((pv_index(n - NVF) + 1.9 + 0.02 * k, op))
        g.phase(pv_mfmas(y), fillers)
        I("s_branch LL_EPILOGUE")
    # ================= epilogue =================
    g.L("LL_EPILOGUE")
    I("s_nop 15")
    I("s_nop 15")
    pstamp(g, 5)
    if DIAG and not DIAG_PRO:
        stamp(g)
        I(f"s_sub_u32 {sreg(S_ACC + 6)}, {sreg(S_TS)}, s72")
        I(f"s_subb_u32 {sreg(S_ACC + 7)}, {sreg(S_TS + 1)}, s73")
        I(f"s_lshl_b32 {sreg(S_T0)}, {sreg(S_WG)}, 2")
        I(f"s_add_u32 {sreg(S_T0)}, {sreg(S_T0)}, {sreg(S_WAVE)}")
        I(f"s_lshl_b32 {sreg(S_T0)}, {sreg(S_T0)}, 5")
        I(f"v_mov_b32 {vreg(V_T)}, {sreg(S_T0)}")
        for k in range(8):
            I(f"v_mov_b32 {vreg(V_T + 2 + k)}, {sreg(S_ACC + k)}")
        I(f"v_cmp_eq_u32_e64 {sreg(S_M0, 2)}, {vreg(V_LANE)}, 0")
        I(f"s_mov_b64 exec, {sreg(S_M0, 2)}")
        I(f"global_store_dwordx4 {vreg(V_T)}, {vreg(V_T + 2, 4)}, {sreg(S_DBG, 2)}")
        I(f"global_store_dwordx4 {vreg(V_T)}, {vreg(V_T + 6, 4)}, {sreg(S_DBG, 2)} offset:16")
        I("s_mov_b64 exec, -1")
    gen_epilogue(g)
    pstamp(g, 6)
    I("s_waitcnt vmcnt(0)")
    pstamp(g, 7)
    if DIAG and DIAG_PRO:                             # the eight stamps -> dbg + 64 * (4 * workgroup + wave)
        I(f"s_lshl_b32 {sreg(S_T0)}, {sreg(S_WG)}, 2")
        I(f"s_add_u32 {sreg(S_T0)}, {sreg(S_T0)}, {sreg(S_WAVE)}")
        I(f"s_lshl_b32 {sreg(S_T0)}, {sreg(S_T0)}, 6")
        I(f"v_mov_b32 {vreg(V_T)}, {sreg(S_T0)}")
        I(f"v_cmp_eq_u32_e64 {sreg(S_M0, 2)}, {vreg(V_LANE)}, 0")
        I(f"s_mov_b64 exec, {sreg(S_M0, 2)}")
        for q4 in range(4):
            for k in range(4):
                I(f"v_mov_b32 {vreg(V_T + 2 + k)}, {sreg(64 + 4 * q4 + k)}")
            I(f"global_store_dwordx4 {vreg(V_T)}, {vreg(V_T + 2, 4)}, {sreg(S_DBG, 2)} offset:{16 * q4}")
        I("s_mov_b64 exec, -1")
        I("s_waitcnt vmcnt(0)")
    I("s_endpgm")
    # ================= idle waves =================
    g.L("LL_IDLE")
    I("s_waitcnt vmcnt(0)")
    I("s_barrier")                                    # (1)
    I("s_barrier")                                    # (2)
    I(f"s_mov_b32 {sreg(S_I)}, 1")
    g.L("LL_IDLE_LOOP")
    for u in (1, 2, 3, 0):
        for op in dma_ops(u + 3, u + 2):
            I(op)
        I("s_waitcnt vmcnt(8)")
        I("s_barrier")
        I(f"s_add_u32 {sreg(S_I)}, {sreg(S_I)}, 1")
        I(f"s_cmp_ge_u32 {sreg(S_I)}, {sreg(S_NT)}")
        I("s_cbranch_scc1 LL_IDLE_END")
    I("s_branch LL_IDLE_LOOP")
    g.L("LL_IDLE_END")
    I("s_waitcnt vmcnt(0)")
    I("s_endpgm")
    return finalize(g.out).replace("LL_", PFX + "_")


# ---- QNORM prologue ------------------------------------------------------------------------------------------------------
QN_W, QN_SS, QN_RI = 64, 96, 128             # v[64:95] norm weight (packed bf16, the lane's 8 channels of each k-step), v[96:127] the planes' sums
                                             # per q-block (16 each), v[128:131] (rinv, rinv) per q-block -- all inside the score buffers / -m tiles,
                                             # which are first written by tile 0


def gen_qnorm_loads(g: Gen):
    """after the Q loads: the head's norm weight (the lane's channels 16 ks + 8 h + (0..7)) and the row's sums of squares, one per
    n-tile plane of the projection (planes past S_NPART are not read: their registers stay +0)"""
    I = g.I
    I(f"v_lshlrev_b32 {vreg(V_T + 4)}, 4, {vreg(V_H)}")                       # 16 h bytes
    for ks in range(8):
        I(f"global_load_dwordx4 {vreg(QN_W + 4 * ks, 4)}, {vreg(V_T + 4)}, {sreg(S_NW, 2)} offset:{32 * ks}")
    for k in range(2 * NPART_MAX):
        I(f"v_mov_b32 {vreg(QN_SS + k)}, 0")
    I(f"s_mov_b64 {sreg(S_T2, 2)}, {sreg(S_SSQ, 2)}")                          # running plane base (s[38:39])
    for j in range(NPART_MAX):
        I(f"s_cmp_ge_u32 {j}, {sreg(S_NPART)}")
        I(f"s_cbranch_scc1 {PFX}_QN_LOADED")
        for qb in range(2):
            I(f"global_load_dword {vreg(QN_SS + 16 * qb + j)}, {vreg(V_T + 2 + qb)}, {sreg(S_T2, 2)}")
        I(f"s_add_u32 {sreg(S_T2)}, {sreg(S_T2)}, {sreg(S_SSQ_STRIDE)}")
        I(f"s_addc_u32 {sreg(S_T3)}, {sreg(S_T3)}, 0")
    g.L(f"{PFX}_QN_LOADED")


def gen_qnorm_convert(g: Gen):
    """q <- bf16( bf16( bf16(x * rinv) * w ) * c ) with rinv = rsq(sum / C + eps): WanRMSNorm's rounding points (x.float() * rsqrt(...)
    rounded to bf16, times the bf16 weight, rounded) followed by this kernel's pre-scaling by c = scale * log2(e)."""
    I = g.I
    for qb in range(2):
        b = QN_SS + 16 * qb
        for j in range(1, NPART_MAX):                                          # plane order, fixed
            I(f"v_add_f32 {vreg(b)}, {vreg(b)}, {vreg(b + j)}")
        I(f"v_mov_b32 {vreg(QN_RI + 2 * qb + 1)}, {sreg(S_EPS)}")
        I(f"v_fma_f32 {vreg(b)}, {vreg(b)}, {sreg(S_INVC)}, {vreg(QN_RI + 2 * qb + 1)}")
        I(f"v_rsq_f32 {vreg(QN_RI + 2 * qb)}, {vreg(b)}")
        I("s_nop 0")
        I(f"v_mov_b32 {vreg(QN_RI + 2 * qb + 1)}, {vreg(QN_RI + 2 * qb)}")
    I(f"s_mov_b32 {sreg(S_C2)}, {sreg(S_C)}")
    I(f"s_mov_b32 {sreg(S_C2 + 1)}, {sreg(S_C)}")
    for qb in range(2):
        for ks in range(8):
            for j in range(4):
                src, wsrc = 32 * qb + 4 * ks + j, QN_W + 4 * ks + j
                t = V_T + 4 + 4 * (j & 1)                                      # two temporaries sets, alternating
                lo, hi, wl, wh = t, t + 1, t + 2, t + 3
                I(f"v_lshlrev_b32 {vreg(lo)}, 16, {vreg(src)}")
                I(f"v_and_b32 {vreg(hi)}, 0xffff0000, {vreg(src)}")
                I(f"v_pk_mul_f32 {vreg(lo, 2)}, {vreg(lo, 2)}, {vreg(QN_RI + 2 * qb, 2)}")
                I(f"v_cvt_pk_bf16_f32 {vreg(lo)}, {vreg(lo)}, {vreg(hi)}")
                I(f"v_and_b32 {vreg(hi)}, 0xffff0000, {vreg(lo)}")
                I(f"v_lshlrev_b32 {vreg(lo)}, 16, {vreg(lo)}")
                I(f"v_lshlrev_b32 {vreg(wl)}, 16, {vreg(wsrc)}")
                I(f"v_and_b32 {vreg(wh)}, 0xffff0000, {vreg(wsrc)}")
                I(f"v_pk_mul_f32 {vreg(lo, 2)}, {vreg(lo, 2)}, {vreg(wl, 2)}")
                I(f"v_cvt_pk_bf16_f32 {vreg(lo)}, {vreg(lo)}, {vreg(hi)}")
                I(f"v_and_b32 {vreg(hi)}, 0xffff0000, {vreg(lo)}")
                I(f"v_lshlrev_b32 {vreg(lo)}, 16, {vreg(lo)}")
                I(f"v_pk_mul_f32 {vreg(lo, 2)}, {sreg(S_C2, 2)}, {vreg(lo, 2)}")
                I(f"v_cvt_pk_bf16_f32 {vreg(lo)}, {vreg(lo)}, {vreg(hi)}")
                I(f"v_accvgpr_write_b32 {areg(A_Q + 32 * qb + 4 * ks + j)}, {vreg(lo)}")


def rescale_rets(x):
    return [f"LL_RESC_RET_{u}" for u in ((0, 2) if x == 0 else (1, 3))]


def mask_rets(x):
    return [f"LL_MASK_RET_{u}" for u in ((0, 2) if x == 0 else (1, 3))]


def pstamp(g: Gen, k: int):
    """timeline stamp k (DIAG_PRO builds only) into s[64 + 2k : 65 + 2k]"""
    if DIAG and DIAG_PRO:
        g.I(f"s_memrealtime {sreg(64 + 2 * k, 2)}")
        g.I("s_waitcnt lgkmcnt(0)")


def stamp(g: Gen, acc=None):
    """diagnostic: s[62:63] = now; if acc: acc += now - last; last = now"""
    if not DIAG or DIAG_PRO:
        return
    g.I("s_memtime s[62:63]")
    g.I("s_waitcnt lgkmcnt(0)")
    if acc is not None:
        g.I(f"s_sub_u32 {sreg(S_T2)}, s62, {sreg(S_TS)}")
        g.I(f"s_subb_u32 {sreg(S_T3)}, s63, {sreg(S_TS + 1)}")
        g.I(f"s_add_u32 {sreg(acc)}, {sreg(acc)}, {sreg(S_T2)}")
        g.I(f"s_addc_u32 {sreg(acc + 1)}, {sreg(acc + 1)}, {sreg(S_T3)}")
    g.I(f"s_mov_b64 {sreg(S_TS, 2)}, s[62:63]")


def gen_tile(g: Gen, u: int):
    """tile i with i % 4 == u (i >= 1): phases A_i and B_i, the barrier, the rare branches"""
    I = g.I
    x, y = u & 1, (u & 1) ^ 1
    # ---- A_i: S(i) = K(i) Q^T  ||  finish(i-1)  ||  first V(i-1) fragments
    fin = finish_ops(y, with_pos=True)
    span = KNOB("FIN_SPAN", 31.0)                    # finish(i-1) is spread over this many MFMA gaps from the start of A_i; beyond 32 it
    late = [op for t, op in fin if t >= 1000]        # row-sum adds deferred to phase B (ADD_LATE)
    fin = [(t, op) for t, op in fin if t < 1000]
    tmax = max([t for t, _ in fin] + [0])            # runs on into B_i, ahead of the P.V k-steps that consume it (k-step k starts at B gap 8 k)
    placed = [(0.3 + (span - 0.3) * k / max(1, len(fin)), op) for k, (t, op) in enumerate(fin)]      # uniform by instruction index:
                                                     # bursts of 3 exp + 5 VALU in one gap (placement by element) cost 3 % of the tile
    fillers = [(p_, op) for p_, op in placed if p_ < 32]
    fin_b = [(p_ - 32, op) for p_, op in placed if p_ >= 32] + spread(late, KNOB("LATE_LO", 1.3), KNOB("LATE_HI", 30.0))
    vf_pos = []                                       # V(i-1) fragment n: read VF_LEAD gaps ahead of its first MFMA (B-phase index pv_index(n)),
    lead = KNOB("VF_LEAD", 3.1)                      # never before the MFMAs of the fragment whose ring slot it takes over
    for n in range(16):
        pos = 32 + pv_index(n) - lead
        if n >= NVF:
            pos = max(pos, 32 + pv_index(n - NVF) + 1.9)
        vf_pos.append(pos)
        if pos < 32:
            for k, op in enumerate(v_frag_reads(n, u - 1)):
                fillers.append((pos + 0.05 * k, op))
    dma_a = KNOB("DMA_IN_A", 0)
    if dma_a and not KNOB("NO_DMA", 0):
        fillers += spread(dma_ops(u + 3, u + 2), 1.4, 30.5)
    g.phase(qk_mfmas(x), fillers)
    stamp(g, S_ACC)
    # ragged last tile: mask S(i) before its row max is taken
    I(f"s_cmp_lg_u32 {sreg(S_I)}, {sreg(S_MASKI)}")                          # S_MASKI = nt - 1 if the last tile is ragged, else -1
    I(f"s_cbranch_scc1 LL_MASK_RET_{u}")
    I(f"s_mov_b32 {sreg(S_RET)}, {mask_rets(x).index(f'LL_MASK_RET_{u}')}")
    I(f"s_branch LL_MASK_{x}")
    g.L(f"LL_MASK_RET_{u}")
    # ---- B_i: O^T += V(i-1)^T P(i-1)^T + row sums  ||  start(i)  ||  K(i+1) -> AGPRs, V(i-1) fragments NVF..15  ||  LDS-DMA K(i+3), V(i+2)
    fillers = []
    for k, op in enumerate([] if KNOB("NO_KREAD", 0) else k_frag_reads(u + 1)):
        fillers.append((0.5 + k * KNOB("KREAD_STEP", 0.95), op))
    for n in range(16):
        if vf_pos[n] >= 32:
            for k, op in enumerate(v_frag_reads(n, u - 1)):
                fillers.append((vf_pos[n] - 32 + 0.02 * k, op))
    fillers += spread(start_ops(x)[(-3 if KNOB("NO_START", 0) else 0):], 3.2, KNOB("START_END", 27.0 if LSUM else 22.0))
    if not dma_a and not KNOB("NO_DMA", 0):
        fillers += spread(dma_ops(u + 3, u + 2), 20.4 if LSUM else 16.4, 39.5 if LSUM else 31.5)
    fillers += fin_b
    g.phase(pv_mfmas(y), fillers)
    # rescale decision (after every P.V of tile i-1 has issued), then the tile barrier
    I(f"s_or_b64 {sreg(S_M0, 2)}, {sreg(S_M0, 2)}, {sreg(S_M1, 2)}")
    I(f"s_cbranch_scc0 LL_RESC_RET_{u}")
    I(f"s_mov_b32 {sreg(S_RET)}, {rescale_rets(x).index(f'LL_RESC_RET_{u}')}")
    I(f"s_branch LL_RESCALE_{x}")
    g.L(f"LL_RESC_RET_{u}")
    stamp(g, S_ACC + 2)
    I("s_waitcnt vmcnt(8)")                           # everything but this tile's 8 pieces has landed: K(i+2), V(i) are in LDS
    I("s_barrier")
    stamp(g, S_ACC + 4)


def gen_epilogue(g: Gen):
    """O^T / l -> bf16 rows.  A lane (r, h) holds, for its query, d = 32 db + 8 g4 + 4 h + (0..3) in registers 4 g4 .. 4 g4 + 3 of
    o[qb][db].  v_permlane32_swap of the packed words of g4 = k (vdst) and k + 1 (src) gives every lane 16 contiguous bytes
    (d = 32 db + 8 (k + h) .. +7): 8 x 16-byte stores per q-block (T21)."""
    I = g.I
    if LSUM:
        for qb in range(2):                           # every register of LACC[qb] holds the lane's query's row sum: take register 0
            I(f"v_rcp_f32 {vreg(V_LACC + 16 * qb)}, {vreg(V_LACC + 16 * qb)}")
    else:
        for qb in range(2):
            l = V_L + 4 * qb
            I(f"v_add_f32 {vreg(l)}, {vreg(l)}, {vreg(l + 1)}")
            I(f"v_add_f32 {vreg(l + 2)}, {vreg(l + 2)}, {vreg(l + 3)}")
        for qb in range(2):
            I(f"v_add_f32 {vreg(V_L + 4 * qb)}, {vreg(V_L + 4 * qb)}, {vreg(V_L + 4 * qb + 2)}")
        for qb in range(2):
            I(f"v_mov_b32 {vreg(V_T + qb)}, {vreg(V_L + 4 * qb)}")
        I("s_nop 0")
        for qb in range(2):                           # the other half of the wave holds the other 32 keys of every tile
            I(f"v_permlane32_swap_b32 {vreg(V_L + 4 * qb)}, {vreg(V_T + qb)}")
        for qb in range(2):
            I(f"v_add_f32 {vreg(V_L + 4 * qb)}, {vreg(V_L + 4 * qb)}, {vreg(V_T + qb)}")
            I(f"v_rcp_f32 {vreg(V_L + 4 * qb)}, {vreg(V_L + 4 * qb)}")
    for qb in range(2):
        inv = (V_LACC + 16 * qb) if LSUM else (V_L + 4 * qb)
        # output row address: O base + row * ldo + (4 h elements -> the swap moves it to 8 (k + h)) ; 64-bit
        I(f"v_mul_lo_u32 {vreg(V_T + 2)}, {vreg(V_ROW + qb)}, {sreg(S_LDO)}")
        I(f"v_lshl_add_u32 {vreg(V_T + 2)}, {vreg(V_H)}, 4, {vreg(V_T + 2)}")            # + 16 h bytes
        I(f"v_cmp_lt_u32_e64 {sreg(S_M0, 2)}, {vreg(V_ROW + qb)}, {sreg(S_ROWS)}")
        for db in range(4):
            for kp in (0, 2):
                w = [V_T + 4, V_T + 5, V_T + 6, V_T + 7]            # ax, ay, bx, by
                for half, g4 in enumerate((kp, kp + 1)):
                    for pr in range(2):
                        a0 = A_O + 64 * qb + 16 * db + 4 * g4 + 2 * pr
                        I(f"v_accvgpr_read_b32 {vreg(V_T + 8)}, {areg(a0)}")
                        I(f"v_accvgpr_read_b32 {vreg(V_T + 9)}, {areg(a0 + 1)}")
                        I(f"v_mul_f32 {vreg(V_T + 8)}, {vreg(V_T + 8)}, {vreg(inv)}")
                        I(f"v_mul_f32 {vreg(V_T + 9)}, {vreg(V_T + 9)}, {vreg(inv)}")
                        I(f"v_cvt_pk_bf16_f32 {vreg(w[2 * half + pr])}, {vreg(V_T + 8)}, {vreg(V_T + 9)}")
                I("s_nop 1")
                I(f"v_permlane32_swap_b32 {vreg(w[0])}, {vreg(w[2])}")
                I(f"v_permlane32_swap_b32 {vreg(w[1])}, {vreg(w[3])}")
                # lanes 0-31: [ax ay | upper half's ax ay] -> after the swap w = (ax, ay, bx, by) holds 16 contiguous bytes
                I(f"v_mov_b32 {vreg(V_T + 10)}, {vreg(w[0])}")
                I(f"v_mov_b32 {vreg(V_T + 11)}, {vreg(w[1])}")
                I(f"v_mov_b32 {vreg(V_T + 12)}, {vreg(w[2])}")
                I(f"v_mov_b32 {vreg(V_T + 13)}, {vreg(w[3])}")
                I(f"s_mov_b64 exec, {sreg(S_M0, 2)}")
                I(f"global_store_dwordx4 {vreg(V_T + 2)}, {vreg(V_T + 10, 4)}, {sreg(S_O, 2)} offset:{64 * db + 16 * kp}")
                I("s_mov_b64 exec, -1")


def f32bits(x):
    import struct
    return struct.unpack("<I", struct.pack("<f", x))[0]


# ---- post passes ------------------------------------------------------------------------------------------------------
_RR = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(tok):
    out = set()
    for m in _RR.finditer(tok):
        if m.group(1):
            out |= {(m.group(1), k) for k in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def split_ops(line):
    parts = line.split(None, 1)
    mn = parts[0]
    ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
    if ops:
        ops[-1] = ops[-1].split()[0] if ops[-1].split() else ops[-1]
    return mn, ops


def finalize(lines):
    """Inserts the LDS waits: a register that an un-retired ds_read will write may not be read (or written) before an
    s_waitcnt lgkmcnt(N) with N = number of LDS operations issued after that read.  LDS operations return in order."""
    out = []
    pending = []                                      # destination register sets of un-retired LDS reads, oldest first
    for line in lines:
        if line.endswith(":"):
            if pending:
                raise RuntimeError(f"LDS reads in flight across label {line}")
            out.append(line)
            continue
        mn, ops = split_ops(line)
        if mn in ("s_barrier", "s_endpgm") or mn.startswith("s_cbranch") or mn == "s_branch":
            if pending:
                out.append("s_waitcnt lgkmcnt(0)")
                pending = []
            out.append(line)
            continue
        if mn == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                keep = int(m.group(1))
                pending = pending[len(pending) - keep:] if keep else []
            out.append(line)
            continue
        touched = set()
        for t in ops:
            touched |= regs_of(t)
        need = None
        for idx, dst in enumerate(pending):
            if dst & touched:
                need = idx
        if need is not None:
            keep = min(len(pending) - 1 - need, 15)   # lgkmcnt is a 4-bit field: 15 is the weakest wait there is
            out.append(f"s_waitcnt lgkmcnt({keep})")
            pending = pending[len(pending) - keep:] if keep else []
        if mn.startswith("ds_read"):
            pending.append(regs_of(ops[0]))
        out.append(line)
    fixed = []
    for line in out:                                  # SALU write of m0 -> LDS-DMA needs one wait state
        if fixed and line.startswith("buffer_load") and " lds" in line and re.match(r"s_\w+ m0,", fixed[-1]):
            fixed.append("s_nop 0")
        fixed.append(line)
    return "\n".join(fixed) + "\n"


def lint(text):
    """Software-visible hazards this generator can create (wait states counted in instructions, s_nop N = N + 1):
         MFMA 32x32x16 result -> read or overwritten by a non-MFMA instruction, or read as A/B by an MFMA: >= 12 (hipcc pads s_nop 11)
         VALU write of a VGPR -> MFMA reads it as A / B / C: >= 2
         VALU write -> v_permlane*_swap reads it: >= 2
         SALU write of m0 -> LDS-DMA: >= 1
         transcendental (v_exp / v_rcp / ...) result -> read by the next VALU / MFMA instruction: >= 1
    Straight-line approximation: branches and labels reset nothing (distances only grow across them on the paths taken here)."""
    problems = []
    hist = []                                         # (states_since, kind, regs)
    lines = [l for l in text.splitlines() if l and not l.endswith(":")]
    for ln, line in enumerate(lines):
        mn, ops = split_ops(line)
        states = 1
        if mn == "s_nop":
            states = int(ops[0]) + 1
        is_mfma = mn.startswith("v_mfma")
        is_valu = mn.startswith("v_") and not is_mfma
        reads = set()
        writes = set()
        if mn.startswith("v_") or mn.startswith("ds_") or mn.startswith("global_") or mn.startswith("buffer_"):
            if mn.startswith("global_store") or mn.startswith("ds_write") or (mn.startswith("buffer_load") and "lds" in line):
                for t in ops:
                    reads |= regs_of(t)
            elif mn in ("v_permlane32_swap_b32", "v_permlane16_swap_b32"):
                writes = regs_of(ops[0]) | regs_of(ops[1])
                reads = set(writes)
            else:
                writes = regs_of(ops[0]) if ops else set()
                for t in ops[1:]:
                    reads |= regs_of(t)
        for dist, kind, regs, src in hist:
            if kind == "mfma_d":
                ab = set()
                if is_mfma:
                    ab = regs_of(ops[1]) | regs_of(ops[2])
                    bad = regs & ab
                else:
                    bad = regs & (reads | writes)
                if bad and dist < 12:
                    problems.append(f"{ln}: `{line}` touches MFMA result of `{src}` after {dist} states")
            elif kind == "valu_w":
                if is_mfma and (regs & reads) and dist < 2:
                    problems.append(f"{ln}: `{line}` reads VALU result of `{src}` after {dist} states")
                if mn.startswith("v_permlane") and (regs & reads) and dist < 2:
                    problems.append(f"{ln}: `{line}` swaps VALU result of `{src}` after {dist} states")
            elif kind == "trans_w":
                if (is_valu or is_mfma) and (regs & reads) and dist < 1:
                    problems.append(f"{ln}: `{line}` reads the transcendental result of `{src}` in the next instruction (1 wait state needed)")
            elif kind == "m0_w":
                if "lds" in line and mn.startswith("buffer_load") and dist < 1:
                    problems.append(f"{ln}: `{line}` right after m0 write")
        hist = [(d + states, k, r, s) for d, k, r, s in hist if d + states < 16]
        if is_mfma:
            hist.append((0, "mfma_d", regs_of(ops[0]), line))
        elif is_valu and writes:
            hist.append((0, "valu_w", writes, line))
            if mn.split("_e")[0] in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32"):
                hist.append((0, "trans_w", writes, line))     # gfx940+: trans result -> VALU use needs 1 wait state (hipcc pads s_nop 0)
        if mn.startswith("s_") and ops and ops[0] == "m0":
            hist.append((0, "m0_w", set(), line))
    return problems


def to_inc(text):
    return "".join('"' + l.replace("\\", "\\\\").replace('"', '\\"') + '\\n\\t"\n' for l in text.splitlines())


if __name__ == "__main__":
    mode = "buffer"
    diag = "--diag" in sys.argv
    if diag:
        sys.argv.remove("--diag")
    if "--knobs-header" in sys.argv:
        knobs_header(sys.argv[sys.argv.index("--knobs-header") + 1])
        sys.exit(0)
    if "--dma" in sys.argv:
        k = sys.argv.index("--dma")
        mode = sys.argv[k + 1]
        del sys.argv[k:k + 2]
    qnorm = "--qnorm" in sys.argv
    if qnorm:
        sys.argv.remove("--qnorm")
    txt = generate(mode, "LL" + mode[0].upper() + ("D" if diag else "") + ("N" if qnorm else ""), diag, qnorm)
    probs = lint(txt)
    for p in probs[:40]:
        print("LINT:", p, file=sys.stderr)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(to_inc(txt))
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    n = sum(1 for l in txt.splitlines() if l and not l.endswith(":"))
    print(f"{n} instructions, {len(probs)} lint findings", file=sys.stderr)
    sys.exit(1 if probs else 0)
