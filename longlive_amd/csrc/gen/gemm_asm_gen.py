#!/usr/bin/env python3
"""Generator of the hand-scheduled bf16 GEMM kernels for gfx950 (gemm_asm_kernel_*, longlive_amd/csrc/gemm_asm.hip):
Y[M, N] = epilogue(X[M, K] . W[N, K]^T + bias), the block linears of wan/modules/causal_model.py:406-408 (ffn.0 + GELU),
:456,467 (ffn.2 / o + gate-residual), model.py:172,192-193 (cross o / q).

Same construction as the attention kernel (gen/attn_asm_gen.py: one asm statement of generated text, one wave per SIMD owning
the 512-register file, fillers placed between MFMAs, LDS waits by a dependency pass, hazard linter, executed by
tools/gfx950_emu.py in the CPU suite before any GPU run):

  * workgroup = 4 waves = a 256 (M) x WN (N) output tile; wave w owns rows 64 w .. 64 w + 63 and ALL WN columns:
    accumulators acc[mb][nb] (mb < 2, nb < WN / 32) = 2 WN / 32 tiles of 32 x 32 in AGPRs (224 registers at WN = 224).
    Operands are swapped as in gemm.hip (A := W rows = output columns, B := X rows): a lane then owns ONE output row
    (m = lane & 31) and, per 32-column block, columns 8 g + 4 h + (0..3) -- 8-byte runs that one v_permlane32_swap per pair
    turns into 16-byte stores (T21).
  * K-step BK = 32 (two MFMA k-steps): fragments of step k + 1 are read from LDS into the second half of a double-buffered
    fragment file (WN / 32 + 2 ds_read_b128 per k-step and wave) while the MFMAs of step k run from the first half; every W
    fragment feeds two MFMAs, every X fragment WN / 32.
  * LDS: 4 slots x 32 KiB ([256 X rows | 256 W rows] x 64 B, XOR-swizzled on the SOURCE address: chunk ^ ((row >> 2) & 3)).
    Because the fragments of step k are read during step k - 1, slot k % 4 is free again at the start of step k: the LDS-DMA of
    step k + 4 goes there (buffer_load ... lds, 4 + 4 pieces of 1 KiB per wave and step; descriptors bound the ROWS -- rows past
    M / N read as zeros -- and the K offset travels in soffset).  One barrier per step, counted vmcnt(16): two steps in flight.
  * epilogues in the same text, rounding points as gemm_common.h (v = bf16(acc + bias); GELU x.sigma(2u) with v_exp / v_rcp;
    x + bf16(v . gate[frame]); x + v), packed-f32 VALU where it halves the instruction count (no MFMA runs beside it).
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_asm_gen import Gen, finalize, lint, sreg, vreg, areg, to_inc, f32bits, spread   # noqa: E402

EPI_BIAS, EPI_GELU, EPI_GATE_RES, EPI_RES = 0, 1, 2, 3

# ---- inputs (pinned by the HIP wrapper) ----------------------------------------------------------------------------
S_X, S_W, S_Y, S_BIAS, S_RES, S_GATE = 8, 10, 12, 14, 16, 18      # 64-bit bases (bytes): X row m0; W row n0; Y / RES at (m0, n0); bias + n0;
                                                                  # gate table e + gate_idx * N + n0 (row stride S_GSTRIDE per frame)
S_LDX, S_LDW, S_LDO = 20, 21, 22                                  # row strides in bytes
S_ROWS, S_COLS, S_NK = 23, 24, 25                                 # valid rows (M - m0), valid columns (N - n0, >= WN on this path), K / 32
S_FLEN, S_GSTRIDE, S_M0 = 26, 27, 28                              # gate: frame_len (rows per frame), bytes between frames' gate rows, m0
# working scalars
S_XRS, S_WRS = 32, 36                                             # descriptors
S_WAVE, S_I, S_KOFF, S_T0, S_T1, S_T2 = 40, 41, 42, 44, 45, 46
S_XM0, S_WM0, S_KMAX = 47, 48, 49
S_MSK = 50                                                        # 64-bit lane masks s[50:51], s[52:53]
V_TID = 0

SLOT = lambda s: 32768 * (s & 3)
X_REG, W_REG = 0, 16384                                           # byte offsets of the X / W regions inside a slot


class Cfg:
    def __init__(self, WN, epi):
        assert WN % 32 == 0 and 64 <= WN <= 256
        self.WN, self.NB, self.MB, self.KS, self.epi = WN, WN // 32, 2, 2, epi
        self.nacc = self.MB * self.NB * 16
        assert self.nacc <= 256
        # VGPR map
        self.FW = 0                                               # FW[p][nb][ks] 4 regs each
        self.FX = 2 * self.NB * self.KS * 4                       # FX[p][mb][ks]
        nxt = self.FX + 2 * self.MB * self.KS * 4
        self.V_WOFF, self.V_XOFF = nxt, nxt + 4                   # [ks] low-slot offsets, [2 + ks] the same + 65536
        self.V_DX, self.V_DW = nxt + 8, nxt + 12                  # DMA source offsets (4 + 4 pieces)
        self.V_LANE, self.V_R, self.V_H, self.V_ROW = nxt + 16, nxt + 17, nxt + 18, nxt + 19      # V_ROW[mb] -> +19, +20
        self.V_T = nxt + 24                                       # temporaries
        assert self.V_T + 72 <= 256, self.V_T

    def acc(self, mb, nb):
        return (mb * self.NB + nb) * 16

    def fw(self, p, nb, ks):
        return self.FW + ((p * self.NB + nb) * self.KS + ks) * 4

    def fx(self, p, mb, ks):
        return self.FX + ((p * self.MB + mb) * self.KS + ks) * 4


def mfmas(c: Cfg, p, first):
    out = []
    for ks in range(c.KS):
        for nb in range(c.NB):
            for mb in range(c.MB):
                a = areg(c.acc(mb, nb), 16)
                cin = "0" if (first and ks == 0) else a
                out.append(f"v_mfma_f32_32x32x16_bf16 {a}, {vreg(c.fw(p, nb, ks), 4)}, {vreg(c.fx(p, mb, ks), 4)}, {cin}")
    return out


def frag_reads(c: Cfg, p, slot):
    """fragments of one K-step from ring slot `slot` into fragment buffer p"""
    hi = 2 if SLOT(slot) >= 65536 else 0
    base = SLOT(slot) - (65536 if hi else 0)
    out = []
    for ks in range(c.KS):
        for mb in range(c.MB):
            out.append(f"ds_read_b128 {vreg(c.fx(p, mb, ks), 4)}, {vreg(c.V_XOFF + hi + ks)} offset:{base + X_REG + 2048 * mb}")
        for nb in range(c.NB):
            out.append(f"ds_read_b128 {vreg(c.fw(p, nb, ks), 4)}, {vreg(c.V_WOFF + hi + ks)} offset:{base + W_REG + 2048 * nb}")
    return out


def dma_ops(c: Cfg, slot):
    ops = []
    for i in range(4):
        ops.append(f"s_add_u32 m0, {sreg(S_XM0)}, {SLOT(slot) + X_REG + 1024 * i}")
        ops.append(f"buffer_load_dwordx4 {vreg(c.V_DX + i)}, {sreg(S_XRS, 4)}, {sreg(S_KOFF)} offen lds")
    for i in range(4):
        ops.append(f"s_add_u32 m0, {sreg(S_WM0)}, {SLOT(slot) + W_REG + 1024 * i}")
        ops.append(f"buffer_load_dwordx4 {vreg(c.V_DW + i)}, {sreg(S_WRS, 4)}, {sreg(S_KOFF)} offen lds")
    ops.append(f"s_add_u32 {sreg(S_KOFF)}, {sreg(S_KOFF)}, 64")
    ops.append(f"s_min_u32 {sreg(S_KOFF)}, {sreg(S_KOFF)}, {sreg(S_KMAX)}")       # past the last step: re-stage it (never read) instead of running off K
    return ops


def generate(WN: int, epi: int, prefix: str) -> str:
    c = Cfg(WN, epi)
    g = Gen()
    I = g.I
    T = c.V_T
    # ================= setup =================
    I(f"v_and_b32 {vreg(c.V_LANE)}, 63, {vreg(V_TID)}")
    I(f"v_lshrrev_b32 {vreg(T)}, 6, {vreg(V_TID)}")
    I("s_nop 0")
    I(f"v_readfirstlane_b32 {sreg(S_WAVE)}, {vreg(T)}")
    I(f"v_and_b32 {vreg(c.V_R)}, 31, {vreg(c.V_LANE)}")
    I(f"v_lshrrev_b32 {vreg(c.V_H)}, 5, {vreg(c.V_LANE)}")
    # descriptors: rows bound by num_records (rows past M / N read as zeros), the K offset travels in soffset
    for rs, base, rows, ld in ((S_XRS, S_X, S_ROWS, S_LDX), (S_WRS, S_W, S_COLS, S_LDW)):
        I(f"s_mov_b32 {sreg(rs)}, {sreg(base)}")
        I(f"s_and_b32 {sreg(rs + 1)}, {sreg(base + 1)}, 0xffff")
        I(f"s_min_u32 {sreg(S_T0)}, {sreg(rows)}, 256")
        I(f"s_mul_i32 {sreg(rs + 2)}, {sreg(S_T0)}, {sreg(ld)}")
        I(f"s_mov_b32 {sreg(rs + 3)}, 0x00020000")
    I(f"s_mov_b32 {sreg(S_KOFF)}, 0")
    I(f"s_sub_u32 {sreg(S_KMAX)}, {sreg(S_NK)}, 1")
    I(f"s_lshl_b32 {sreg(S_KMAX)}, {sreg(S_KMAX)}, 6")
    I(f"s_lshl_b32 {sreg(S_XM0)}, {sreg(S_WAVE)}, 12")                     # this wave's pieces: (4 wave + i) KiB into a region
    I(f"s_mov_b32 {sreg(S_WM0)}, {sreg(S_XM0)}")
    # DMA source offsets: piece (4 w + i) = tile rows 16 (4 w + i) + (lane >> 2); the lane at LDS position pos = lane & 3 of its
    # 64-byte row fetches chunk pos ^ ((row >> 2) & 3)
    I(f"v_lshrrev_b32 {vreg(T)}, 2, {vreg(c.V_LANE)}")                     # lane >> 2
    I(f"v_and_b32 {vreg(T + 1)}, 3, {vreg(c.V_LANE)}")                     # pos
    I(f"s_lshl_b32 {sreg(S_T0)}, {sreg(S_WAVE)}, 6")                       # 64 w
    for i in range(4):
        I(f"v_add_u32 {vreg(T + 2)}, {sreg(S_T0)}, {vreg(T)}")
        I(f"v_add_u32 {vreg(T + 2)}, {16 * i}, {vreg(T + 2)}")             # row within the 256-row region
        I(f"v_lshrrev_b32 {vreg(T + 3)}, 2, {vreg(T + 2)}")
        I(f"v_and_b32 {vreg(T + 3)}, 3, {vreg(T + 3)}")
        I(f"v_xor_b32 {vreg(T + 3)}, {vreg(T + 3)}, {vreg(T + 1)}")
        I(f"v_lshlrev_b32 {vreg(T + 3)}, 4, {vreg(T + 3)}")
        I(f"v_mul_lo_u32 {vreg(c.V_DX + i)}, {vreg(T + 2)}, {sreg(S_LDX)}")
        I(f"v_add_u32 {vreg(c.V_DX + i)}, {vreg(c.V_DX + i)}, {vreg(T + 3)}")
        I(f"v_mul_lo_u32 {vreg(c.V_DW + i)}, {vreg(T + 2)}, {sreg(S_LDW)}")
        I(f"v_add_u32 {vreg(c.V_DW + i)}, {vreg(c.V_DW + i)}, {vreg(T + 3)}")
    # prologue staging: steps 0..3 into slots 0..3
    for st in range(4):
        for op in dma_ops(c, st):
            I(op)
    # idle waves (their 64 rows are all past M) only stage and synchronise
    I(f"s_cmp_ge_u32 {sreg(S_T0)}, {sreg(S_ROWS)}")
    I(f"s_cbranch_scc1 {prefix}_IDLE")
    # fragment read offsets: row r (+ 32 blocks by immediate), chunk (2 ks + h) ^ ((r >> 2) & 3); X rows start at 64 w
    I(f"v_lshrrev_b32 {vreg(T)}, 2, {vreg(c.V_R)}")
    I(f"v_and_b32 {vreg(T)}, 3, {vreg(T)}")
    I(f"v_lshlrev_b32 {vreg(T + 1)}, 6, {vreg(c.V_R)}")                    # r * 64
    I(f"s_lshl_b32 {sreg(S_T1)}, {sreg(S_WAVE)}, 12")                      # 64 w rows * 64 B
    for ks in range(c.KS):
        I(f"v_add_u32 {vreg(T + 2)}, {2 * ks}, {vreg(c.V_H)}")
        I(f"v_xor_b32 {vreg(T + 2)}, {vreg(T + 2)}, {vreg(T)}")
        I(f"v_lshl_add_u32 {vreg(c.V_WOFF + ks)}, {vreg(T + 2)}, 4, {vreg(T + 1)}")
        I(f"v_add_u32 {vreg(c.V_XOFF + ks)}, {sreg(S_T1)}, {vreg(c.V_WOFF + ks)}")
        I(f"v_add_u32 {vreg(c.V_WOFF + 2 + ks)}, 0x10000, {vreg(c.V_WOFF + ks)}")
        I(f"v_add_u32 {vreg(c.V_XOFF + 2 + ks)}, 0x10000, {vreg(c.V_XOFF + ks)}")
    for mb in range(c.MB):
        I(f"v_add_u32 {vreg(c.V_ROW + mb)}, {sreg(S_T0)}, {vreg(c.V_R)}")
        if mb:
            I(f"v_add_u32 {vreg(c.V_ROW + mb)}, {32 * mb}, {vreg(c.V_ROW + mb)}")
    for r in range(c.nacc):                            # accumulators start from zero (while the first tiles fly)
        I(f"v_accvgpr_write_b32 {areg(r)}, 0")
    I("s_waitcnt vmcnt(24)")                           # step 0 has landed (3 later steps = 24 pieces may fly)
    I("s_barrier")                                     # (1)
    for op in frag_reads(c, 0, 0):
        I(op)
    I("s_waitcnt vmcnt(16)")                           # step 1 has landed
    I("s_barrier")                                     # (2) every wave has read slot 0: step 4 may be staged into it
    I(f"s_mov_b32 {sreg(S_I)}, 0")
    # ================= main loop: step k computes from fragment buffer k & 1, reads step k + 1 (slot (k + 1) & 3) into the other,
    # stages step k + 4 into slot k & 3.  Unrolled by 4.
    g.L(f"{prefix}_LOOP")
    for u in range(4):
        p = u & 1
        fillers = [(0.6 + k * (len(mfmas(c, p, False)) - 8) / (c.NB * 2 + 4), op) for k, op in enumerate(frag_reads(c, p ^ 1, u + 1))]
        fillers += spread(dma_ops(c, u), 2.3, len(mfmas(c, p, False)) - 1.5)
        g.phase(mfmas(c, p, False), fillers)
        I("s_waitcnt vmcnt(16)")                       # all but the last two steps' pieces have landed: step k + 2 is in LDS
        I("s_barrier")
        I(f"s_add_u32 {sreg(S_I)}, {sreg(S_I)}, 1")
        I(f"s_cmp_ge_u32 {sreg(S_I)}, {sreg(S_NK)}")
        I(f"s_cbranch_scc1 {prefix}_EPI")
    I(f"s_branch {prefix}_LOOP")
    # ================= epilogue =================
    g.L(f"{prefix}_EPI")
    I("s_nop 15")
    I("s_nop 15")
    gen_epilogue(g, c)
    I("s_waitcnt vmcnt(0)")
    I("s_endpgm")
    # ================= idle waves =================
    g.L(f"{prefix}_IDLE")
    I("s_waitcnt vmcnt(24)")
    I("s_barrier")
    I("s_waitcnt vmcnt(16)")
    I("s_barrier")
    I(f"s_mov_b32 {sreg(S_I)}, 0")
    g.L(f"{prefix}_IDLE_LOOP")
    for u in range(4):
        for op in dma_ops(c, u):
            I(op)
        I("s_waitcnt vmcnt(16)")
        I("s_barrier")
        I(f"s_add_u32 {sreg(S_I)}, {sreg(S_I)}, 1")
        I(f"s_cmp_ge_u32 {sreg(S_I)}, {sreg(S_NK)}")
        I(f"s_cbranch_scc1 {prefix}_IDLE_END")
    I(f"s_branch {prefix}_IDLE_LOOP")
    g.L(f"{prefix}_IDLE_END")
    I("s_waitcnt vmcnt(0)")
    I("s_endpgm")
    return finalize(g.out)


def gen_epilogue(g: Gen, c: Cfg):
    """per (mb, nb): 16 accumulators of a lane = its row m, columns 32 nb + 8 g4 + 4 h + (0..3), g4 = 0..3"""
    I = g.I
    T = c.V_T
    epi = c.epi
    K0, K1, CEXP = 0.7978845608028654, 0.044715, -2.0 * 1.4426950408889634
    # constants in SGPR pairs for the packed ops
    cons = {}
    def const_pair(name, val, s0):
        I(f"s_mov_b32 {sreg(s0)}, {hex(f32bits(val))}")
        I(f"s_mov_b32 {sreg(s0 + 1)}, {hex(f32bits(val))}")
        cons[name] = s0
    if epi == EPI_GELU:
        const_pair("k1", K1, 54); const_pair("k0", K0, 56); const_pair("ce", CEXP, 58); const_pair("one", 1.0, 60)
    # row addressing: byte offset of the lane's row in Y / RES: row * ldo + 16 h (the swap gives the upper half the second 16 bytes... see below)
    for mb in range(c.MB):
        I(f"v_mul_lo_u32 {vreg(T + 60 + mb)}, {vreg(c.V_ROW + mb)}, {sreg(S_LDO)}")
        I(f"v_lshl_add_u32 {vreg(T + 60 + mb)}, {vreg(c.V_H)}, 4, {vreg(T + 60 + mb)}")
    I(f"v_lshlrev_b32 {vreg(T + 62)}, 3, {vreg(c.V_H)}")                     # bias / gate column byte offset of this half: 8 h
    if epi == EPI_GATE_RES:
        # gate row of the lane's row: frame = (m0 + row) / frame_len  (integer division via float with one correction each way)
        for mb in range(c.MB):
            m, q, t = T + 66, T + 64 + mb, T + 48
            I(f"v_add_u32 {vreg(m)}, {sreg(S_M0)}, {vreg(c.V_ROW + mb)}")
            I(f"v_cvt_f32_u32 {vreg(t)}, {vreg(m)}")
            I(f"v_cvt_f32_u32 {vreg(t + 1)}, {sreg(S_FLEN)}")
            I(f"v_rcp_f32 {vreg(t + 1)}, {vreg(t + 1)}")
            I("s_nop 0")
            I(f"v_mul_f32 {vreg(t)}, {vreg(t)}, {vreg(t + 1)}")
            I(f"v_cvt_u32_f32 {vreg(q)}, {vreg(t)}")                         # q ~ m / flen (may be one off)
            I(f"v_mul_lo_u32 {vreg(t)}, {vreg(q)}, {sreg(S_FLEN)}")          # q * flen
            I(f"v_cmp_gt_u32_e64 {sreg(S_MSK, 2)}, {vreg(t)}, {vreg(m)}")    # too big -> q - 1
            I(f"v_cndmask_b32_e64 {vreg(t + 2)}, 0, 1, {sreg(S_MSK, 2)}")
            I(f"v_sub_u32 {vreg(q)}, {vreg(q)}, {vreg(t + 2)}")
            I(f"v_mul_lo_u32 {vreg(t)}, {vreg(q)}, {sreg(S_FLEN)}")
            I(f"v_add_u32 {vreg(t)}, {sreg(S_FLEN)}, {vreg(t)}")             # (q + 1) * flen <= m -> q + 1
            I(f"v_cmp_le_u32_e64 {sreg(S_MSK, 2)}, {vreg(t)}, {vreg(m)}")
            I(f"v_cndmask_b32_e64 {vreg(t + 2)}, 0, 1, {sreg(S_MSK, 2)}")
            I(f"v_add_u32 {vreg(q)}, {vreg(q)}, {vreg(t + 2)}")
            I(f"v_mul_lo_u32 {vreg(q)}, {vreg(q)}, {sreg(S_GSTRIDE)}")       # byte offset of the frame's gate row
            I(f"v_lshl_add_u32 {vreg(q)}, {vreg(c.V_H)}, 3, {vreg(q)}")      # + 8 h: the gate is applied in the accumulator layout (as the bias)
    for mb in range(c.MB):
        I(f"v_cmp_lt_u32_e64 {sreg(S_MSK + 2 * mb, 2)}, {vreg(c.V_ROW + mb)}, {sreg(S_ROWS)}")
    for nb in range(c.NB):
        # bias of this column block: lane needs columns 32 nb + 8 g4 + 4 h + (0..3): 4 loads of 8 bytes
        for g4 in range(4):
            I(f"global_load_dwordx2 {vreg(T + 40 + 2 * g4, 2)}, {vreg(T + 62)}, {sreg(S_BIAS, 2)} offset:{64 * nb + 16 * g4}")
        I("s_waitcnt vmcnt(0)")
        for g4 in range(4):                                                  # bf16 x4 -> f32 x4: T+24+4 g4 .. +3
            for d in range(2):
                src = T + 40 + 2 * g4 + d
                I(f"v_lshlrev_b32 {vreg(T + 24 + 4 * g4 + 2 * d)}, 16, {vreg(src)}")
                I(f"v_and_b32 {vreg(T + 24 + 4 * g4 + 2 * d + 1)}, 0xffff0000, {vreg(src)}")
        for mb in range(c.MB):
            a0 = c.acc(mb, nb)
            # v = bf16(acc + bias), kept as f32 in T+0..15
            for r in range(16):
                I(f"v_accvgpr_read_b32 {vreg(T + r)}, {areg(a0 + r)}")
            for r in range(0, 16, 2):
                I(f"v_pk_add_f32 {vreg(T + r, 2)}, {vreg(T + r, 2)}, {vreg(T + 24 + r, 2)}")
            for r in range(0, 16, 2):                                        # round to bf16 and back
                I(f"v_cvt_pk_bf16_f32 {vreg(T + 16 + r // 2)}, {vreg(T + r)}, {vreg(T + r + 1)}")
            if epi in (EPI_GELU, EPI_GATE_RES):
                for r in range(0, 16, 2):
                    I(f"v_lshlrev_b32 {vreg(T + r)}, 16, {vreg(T + 16 + r // 2)}")
                    I(f"v_and_b32 {vreg(T + r + 1)}, 0xffff0000, {vreg(T + 16 + r // 2)}")
            if epi == EPI_GATE_RES:                                          # w = bf16(v * gate[frame][n]) in the accumulator layout
                for g4 in range(4):
                    I(f"global_load_dwordx2 {vreg(T + 48 + 2 * g4, 2)}, {vreg(T + 64 + mb)}, {sreg(S_GATE, 2)} offset:{64 * nb + 16 * g4}")
                I("s_waitcnt vmcnt(0)")
                for r in range(0, 16, 2):
                    src = T + 48 + r // 2
                    I(f"v_lshlrev_b32 {vreg(T + 56)}, 16, {vreg(src)}")
                    I(f"v_and_b32 {vreg(T + 57)}, 0xffff0000, {vreg(src)}")
                    I(f"v_pk_mul_f32 {vreg(T + r, 2)}, {vreg(T + r, 2)}, {vreg(T + 56, 2)}")
                for r in range(0, 16, 2):
                    I(f"v_cvt_pk_bf16_f32 {vreg(T + 16 + r // 2)}, {vreg(T + r)}, {vreg(T + r + 1)}")
            if epi == EPI_GELU:
                # y = x * rcp(1 + exp2(ce * (k0 * (x + ((k1 * x) * x) * x))))   (gemm_common.h / common.h gelu_tanh, contraction off)
                for r in range(0, 16, 2):
                    x, t = vreg(T + r, 2), vreg(T + 44 + (r % 4), 2)
                    I(f"v_pk_mul_f32 {t}, {sreg(cons['k1'], 2)}, {x}")
                    I(f"v_pk_mul_f32 {t}, {t}, {x}")
                    I(f"v_pk_mul_f32 {t}, {t}, {x}")
                    I(f"v_pk_add_f32 {t}, {x}, {t}")
                    I(f"v_pk_mul_f32 {t}, {sreg(cons['k0'], 2)}, {t}")
                    I(f"v_pk_mul_f32 {t}, {sreg(cons['ce'], 2)}, {t}")
                    I(f"v_exp_f32 {vreg(T + 44 + (r % 4))}, {vreg(T + 44 + (r % 4))}")
                    I(f"v_exp_f32 {vreg(T + 45 + (r % 4))}, {vreg(T + 45 + (r % 4))}")
                    I("s_nop 0")                                             # transcendental result -> VALU: one wait state
                    I(f"v_pk_add_f32 {t}, {sreg(cons['one'], 2)}, {t}")
                    I(f"v_rcp_f32 {vreg(T + 44 + (r % 4))}, {vreg(T + 44 + (r % 4))}")
                    I(f"v_rcp_f32 {vreg(T + 45 + (r % 4))}, {vreg(T + 45 + (r % 4))}")
                    I("s_nop 0")
                    I(f"v_pk_mul_f32 {x}, {x}, {t}")
                for r in range(0, 16, 2):
                    I(f"v_cvt_pk_bf16_f32 {vreg(T + 16 + r // 2)}, {vreg(T + r)}, {vreg(T + r + 1)}")
            # packed bf16 in T+16..23: dwords (2 g4, 2 g4 + 1) = columns 8 g4 + 4 h + (0..3).  T21: swap pairs (g4 = 0,1) and (2,3)
            for k in (0, 2):
                ax, ay, bx, by = T + 16 + 2 * k, T + 17 + 2 * k, T + 18 + 2 * k, T + 19 + 2 * k
                I("s_nop 1")
                I(f"v_permlane32_swap_b32 {vreg(ax)}, {vreg(bx)}")
                I(f"v_permlane32_swap_b32 {vreg(ay)}, {vreg(by)}")
            off_y = 64 * nb
            if epi in (EPI_BIAS, EPI_GELU):
                I(f"s_mov_b64 exec, {sreg(S_MSK + 2 * mb, 2)}")
                for k in (0, 2):
                    I(f"global_store_dwordx4 {vreg(T + 60 + mb)}, {vreg(T + 16 + 2 * k, 4)}, {sreg(S_Y, 2)} offset:{off_y + 16 * k}")
                I("s_mov_b64 exec, -1")
            else:
                # residual arithmetic in the 8-column layout after the swap: out = bf16(res + w), w = v or bf16(v * gate)
                I(f"s_mov_b64 exec, {sreg(S_MSK + 2 * mb, 2)}")
                for k in (0, 2):
                    I(f"global_load_dwordx4 {vreg(T + 40 + 2 * k, 4)}, {vreg(T + 60 + mb)}, {sreg(S_RES, 2)} offset:{off_y + 16 * k}")
                I("s_mov_b64 exec, -1")
                I("s_waitcnt vmcnt(0)")
                for k in (0, 2):
                    for d in range(4):                                       # 8 columns: res + v in f32, rounded once
                        vsrc, rsrc = T + 16 + 2 * k + d, T + 40 + 2 * k + d
                        I(f"v_lshlrev_b32 {vreg(T)}, 16, {vreg(vsrc)}")
                        I(f"v_and_b32 {vreg(T + 1)}, 0xffff0000, {vreg(vsrc)}")
                        I(f"v_lshlrev_b32 {vreg(T + 2)}, 16, {vreg(rsrc)}")
                        I(f"v_and_b32 {vreg(T + 3)}, 0xffff0000, {vreg(rsrc)}")
                        I(f"v_pk_add_f32 {vreg(T, 2)}, {vreg(T + 2, 2)}, {vreg(T, 2)}")
                        I(f"v_cvt_pk_bf16_f32 {vreg(vsrc)}, {vreg(T)}, {vreg(T + 1)}")
                I(f"s_mov_b64 exec, {sreg(S_MSK + 2 * mb, 2)}")
                for k in (0, 2):
                    I(f"global_store_dwordx4 {vreg(T + 60 + mb)}, {vreg(T + 16 + 2 * k, 4)}, {sreg(S_Y, 2)} offset:{off_y + 16 * k}")
                I("s_mov_b64 exec, -1")


if __name__ == "__main__":
    WN, epi = int(sys.argv[1]), int(sys.argv[2])
    txt = generate(WN, epi, f"GA{WN}E{epi}")
    probs = lint(txt)
    for p in probs[:20]:
        print("LINT:", p, file=sys.stderr)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(to_inc(txt))
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(txt)
    n = sum(1 for l in txt.splitlines() if l and not l.endswith(":"))
    print(f"gemm WN={WN} epi={epi}: {n} instructions, {len(probs)} lint findings", file=sys.stderr)
    sys.exit(1 if probs else 0)
