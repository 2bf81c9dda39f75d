#!/usr/bin/env python3
"""Functional emulator for the subset of the gfx950 ISA that the generated attention kernel uses (tools/attn_asm_gen.py).

TEST INFRASTRUCTURE ONLY.  It exists because this build container has no GPU: the hand-scheduled kernel is a few thousand
lines of generated assembly whose register allocation, LDS addressing, pipelining and wait counts have to be right before a
GPU minute is spent on it.  The emulator executes the very text that is assembled into the library, one workgroup at a time
(1-8 waves in lock step between barriers), on numpy, and the CPU test-suite checks the result against an fp64 attention.

What it models:
  * registers: 256 VGPRs + 256 AGPRs x 64 lanes, SGPRs, VCC, EXEC (only all-ones is accepted by cross-lane ops), SCC, M0
  * MFMA lane layouts as documented for gfx950 (cdna_hip_programming.md section 3): v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16
  * LDS (160 KiB) with ds_read_b128 / _b64 / ds_read_b64_tr_b16 / ds_write_*; LDS-DMA (buffer_load_dwordx4 ... lds,
    global_load_lds_dwordx4: destination = M0 base + instruction offset + 16 x lane)
  * asynchronous completion in THREE modes, all of which a correct kernel must survive:
      eager : every memory operation completes at issue          (catches write-after-read races on LDS ring slots)
      lazy  : an operation completes only when an s_waitcnt retires it; until then a register destination holds a NaN poison
              and an LDS-DMA destination keeps its old bytes        (catches missing / under-counted waits)
      mixed : vector-memory operations (LDS-DMA included) complete at issue, LDS reads only when retired (catches an LDS-DMA
              overwriting a ring slot whose fragment reads have been issued but not waited for)
    vmcnt counts VMEM loads, stores and LDS-DMA together in issue order; lgkmcnt counts LDS operations in issue order
  * a static check of software-visible hazards is NOT done here (see attn_asm_gen.lint): the emulator has no notion of wait states.

What it does not model: timing, bank conflicts, caches, denormal modes, rounding-mode bits; v_exp_f32 is exact exp2 in float32.
"""
from __future__ import annotations

import re
import numpy as np

U32 = np.uint32
POISON = np.uint32(0x7FC0DEAD)       # a NaN: any use before the wait shows up as NaN in the output


def f2u(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def u2f(x):
    return np.asarray(x, dtype=np.uint32).view(np.float32)


def bf16_round(x):
    """float32 array -> bf16 bits (uint32 holding 16 bits), round to nearest even, NaN kept quiet."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint32)


def bf16_to_f32(bits):
    return (np.asarray(bits, dtype=np.uint32) << 16).view(np.float32)


class EmuError(Exception):
    pass


class Memory:
    """Flat device address space: named numpy byte buffers at fake, well-separated base addresses."""

    def __init__(self):
        self.bufs = []               # (base, array uint8)
        self.next = 0x7F0000100000

    def alloc(self, arr: np.ndarray) -> int:
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1).copy()
        base = self.next
        self.bufs.append((base, a))
        self.next += ((a.size + 0xFFFFF) & ~0xFFFFF) + 0x100000
        return base

    def get(self, base: int) -> np.ndarray:
        for b, a in self.bufs:
            if b == base:
                return a
        raise KeyError(base)

    def _find(self, addr: int, n: int):
        for b, a in self.bufs:
            if b <= addr and addr + n <= b + a.size:
                return a, addr - b
        raise EmuError(f"global access of {n} bytes at {addr:#x} is outside every buffer")

    def read(self, addr: int, n: int) -> np.ndarray:
        a, o = self._find(int(addr), n)
        return a[o:o + n]

    def write(self, addr: int, data: np.ndarray):
        a, o = self._find(int(addr), data.size)
        a[o:o + data.size] = data


_REG = re.compile(r"^(v|a|s)(\d+)$")
_RANGE = re.compile(r"^(v|a|s)\[(\d+):(\d+)\]$")
SPECIAL = {"vcc_lo": 106, "vcc_hi": 107, "m0": 124, "exec_lo": 126, "exec_hi": 127}
INLINE_F = {"0.5": 0.5, "-0.5": -0.5, "1.0": 1.0, "-1.0": -1.0, "2.0": 2.0, "-2.0": -2.0, "4.0": 4.0, "-4.0": -4.0}


class Op:
    """One parsed operand."""
    __slots__ = ("kind", "idx", "n", "val", "neg", "abs")

    def __init__(self, kind, idx=0, n=1, val=0, neg=False, abs_=False):
        self.kind, self.idx, self.n, self.val, self.neg, self.abs = kind, idx, n, val, neg, abs_

    def __repr__(self):
        return f"Op({self.kind},{self.idx},{self.n},{self.val})"


def parse_operand(t: str) -> Op:
    t = t.strip()
    neg = abs_ = False
    if t.startswith("-") and not re.match(r"^-[\d.]", t):
        neg, t = True, t[1:]
    if t.startswith("|") and t.endswith("|"):
        abs_, t = True, t[1:-1]
    m = _REG.match(t)
    if m:
        return Op(m.group(1), int(m.group(2)), 1, 0, neg, abs_)
    m = _RANGE.match(t)
    if m:
        lo, hi = int(m.group(2)), int(m.group(3))
        return Op(m.group(1), lo, hi - lo + 1, 0, neg, abs_)
    if t in SPECIAL:
        return Op("s", SPECIAL[t], 1)
    if t == "vcc":
        return Op("s", 106, 2)
    if t == "exec":
        return Op("s", 126, 2)
    if t == "scc":
        return Op("scc")
    if t == "off":
        return Op("off")
    if t in INLINE_F:
        return Op("imm", val=int(f2u(INLINE_F[t])))
    if re.match(r"^-?0x[0-9a-fA-F]+$", t):
        return Op("imm", val=int(t, 16) & 0xFFFFFFFF)
    if re.match(r"^-?\d+$", t):
        return Op("imm", val=int(t) & 0xFFFFFFFF)
    if re.match(r"^-?\d*\.\d+(e-?\d+)?$", t) or re.match(r"^-?\d+e-?\d+$", t):
        return Op("imm", val=int(f2u(float(t))))
    return Op("label", val=t)


class Inst:
    __slots__ = ("mn", "ops", "mods", "text", "line")

    def __init__(self, mn, ops, mods, text, line):
        self.mn, self.ops, self.mods, self.text, self.line = mn, ops, mods, text, line


def parse_program(text: str):
    """-> (list[Inst], {label: index}).  Accepts `;` and `//` comments, `label:` lines, operands separated by commas, trailing
    modifiers (`offset:16 offen lds sc1 ...`)."""
    insts, labels = [], {}
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split("//")[0].split(";")[0].strip()
        if not line:
            continue
        while True:
            m = re.match(r"^([A-Za-z_.$][\w.$]*):\s*(.*)$", line)
            if not m:
                break
            labels[m.group(1)] = len(insts)
            line = m.group(2).strip()
        if not line:
            continue
        parts = line.split(None, 1)
        mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        mods = {}
        ops = []
        if mn == "s_waitcnt":
            for key, val in re.findall(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)", rest):
                mods[key] = int(val)
        else:
            toks = [t.strip() for t in rest.split(",")] if rest else []
            if toks:
                last = toks[-1].split()
                if last:
                    toks[-1] = last[0]
                    for extra in last[1:]:
                        if ":" in extra:
                            k, v = extra.split(":", 1)
                            mods[k] = int(v, 0)
                        else:
                            mods[extra] = True
            # a leading bare modifier list such as "off offset:16" after the comma is handled above; `off` is an operand
            ops = [parse_operand(t) for t in toks if t != ""]
        insts.append(Inst(mn, ops, mods, line, ln))
    return insts, labels


class Wave:
    def __init__(self, wid, nlanes=64):
        self.id = wid
        self.v = np.zeros((256, 64), dtype=np.uint32)
        self.a = np.zeros((256, 64), dtype=np.uint32)
        self.s = np.zeros(128, dtype=np.uint32)
        self.s[126] = 0xFFFFFFFF
        self.s[127] = 0xFFFFFFFF
        self.scc = 0
        self.pc = 0
        self.done = False
        self.at_barrier = False
        self.vm = []          # pending VMEM ops (closures), oldest first
        self.lgkm = []        # pending LDS ops
        self.prio = 0
        self.n_exec = 0


class Machine:
    def __init__(self, program_text: str, mem: Memory, nwaves: int, mode: str = "lazy", lds_bytes: int = 160 * 1024):
        assert mode in ("lazy", "eager", "mixed")
        self.insts, self.labels = parse_program(program_text)
        self.mem, self.mode = mem, mode
        self.lds = np.zeros(lds_bytes, dtype=np.uint8)
        self.waves = [Wave(i) for i in range(nwaves)]
        self.trace = None
        self.max_pending = {"vm": 0, "lgkm": 0}

    # ---- operand access ---------------------------------------------------------------------------------------------
    def rs(self, w: Wave, op: Op, n: int = 1) -> int:
        """scalar source as python int (n dwords)"""
        if op.kind == "imm":
            v = op.val
            if n == 2 and v & 0x80000000 and v < 0x100000000:      # sign-extend small negative inline constants
                v |= 0xFFFFFFFF00000000
            return v
        if op.kind == "scc":
            return w.scc
        if op.kind != "s":
            raise EmuError(f"scalar operand expected, got {op}")
        v = 0
        for i in range(max(n, op.n)):
            v |= int(w.s[op.idx + i]) << (32 * i)
        return v

    def ws(self, w: Wave, op: Op, val: int, n: int = 1):
        if op.kind != "s":
            raise EmuError(f"scalar destination expected, got {op}")
        for i in range(max(n, op.n)):
            w.s[op.idx + i] = (val >> (32 * i)) & 0xFFFFFFFF

    def rv(self, w: Wave, op: Op, i: int = 0) -> np.ndarray:
        """vector source dword i as uint32[64] (scalars / immediates broadcast); applies neg/abs as float modifiers"""
        if op.kind == "v":
            x = w.v[op.idx + i]
        elif op.kind == "a":
            x = w.a[op.idx + i]
        elif op.kind == "s":
            x = np.full(64, w.s[op.idx + i], dtype=np.uint32)
        elif op.kind == "imm":
            x = np.full(64, op.val, dtype=np.uint32)
        else:
            raise EmuError(f"vector source expected, got {op}")
        if op.abs:
            x = x & U32(0x7FFFFFFF)
        if op.neg:
            x = x ^ U32(0x80000000)
        return x

    def wv(self, w: Wave, op: Op, val, i: int = 0):
        val = np.asarray(val).astype(np.uint32)
        if op.kind == "v":
            w.v[op.idx + i] = val
        elif op.kind == "a":
            w.a[op.idx + i] = val
        else:
            raise EmuError(f"vector destination expected, got {op}")

    def exec_mask(self, w: Wave) -> np.ndarray:
        e = int(w.s[126]) | (int(w.s[127]) << 32)
        return np.array([(e >> i) & 1 for i in range(64)], dtype=bool)

    def full_exec(self, w: Wave, what: str):
        if int(w.s[126]) != 0xFFFFFFFF or int(w.s[127]) != 0xFFFFFFFF:
            raise EmuError(f"{what} requires EXEC = all ones")

    # ---- async queues -----------------------------------------------------------------------------------------------
    def _issue(self, w: Wave, q: str, fn):
        if self.mode == "eager" or (self.mode == "mixed" and q == "vm"):
            fn()
            getattr(w, q).append(None)
        else:
            getattr(w, q).append(fn)
        self.max_pending[q] = max(self.max_pending[q], len(getattr(w, q)))

    def _retire(self, w: Wave, q: str, keep: int):
        lst = getattr(w, q)
        while len(lst) > keep:
            fn = lst.pop(0)
            if fn is not None:
                fn()

    # ---- execution --------------------------------------------------------------------------------------------------
    def run(self, max_steps: int = 50_000_000):
        steps = 0
        while True:
            progressed = False
            for w in self.waves:
                while not w.done and not w.at_barrier:
                    self.step(w)
                    progressed = True
                    steps += 1
                    if steps > max_steps:
                        raise EmuError("step limit exceeded (endless loop?)")
            live = [w for w in self.waves if not w.done]
            if not live:
                break
            if all(w.at_barrier for w in live):
                for w in live:
                    w.at_barrier = False
                continue
            if not progressed:
                raise EmuError("deadlock: some waves ended while others wait at a barrier")
        for w in self.waves:            # outstanding stores complete at program end
            self._retire(w, "vm", 0)
            self._retire(w, "lgkm", 0)
        return steps

    def step(self, w: Wave):
        if w.pc >= len(self.insts):
            raise EmuError(f"wave {w.id} ran off the end of the program")
        ins = self.insts[w.pc]
        w.pc += 1
        w.n_exec += 1
        fn = getattr(self, "i_" + ins.mn.replace(".", "_"), None)
        if fn is None:
            base = re.sub(r"_e32$|_e64$", "", ins.mn)
            fn = getattr(self, "i_" + base, None)
        if fn is None:
            raise EmuError(f"line {ins.line}: unsupported instruction `{ins.text}`")
        try:
            fn(w, ins)
        except EmuError as e:
            raise EmuError(f"line {ins.line} `{ins.text}` (wave {w.id}): {e}") from None

    # ---- SALU -------------------------------------------------------------------------------------------------------
    def i_s_nop(self, w, i): pass
    def i_s_setprio(self, w, i): pass
    def i_s_sleep(self, w, i): pass

    def i_s_endpgm(self, w, i):
        w.done = True

    def i_s_barrier(self, w, i):
        w.at_barrier = True

    def i_s_waitcnt(self, w, i):
        if "vmcnt" in i.mods:
            self._retire(w, "vm", i.mods["vmcnt"])
        if "lgkmcnt" in i.mods:
            self._retire(w, "lgkm", i.mods["lgkmcnt"])

    def i_s_memtime(self, w, i):
        self.ws(w, i.ops[0], w.n_exec * 4, 2)

    def i_s_subb_u32(self, w, i):
        c = w.scc
        a, b, r = self._sarith(w, i, lambda a, b: a - b - c)
        w.scc = 1 if b + c > a else 0

    def i_s_mov_b32(self, w, i): self.ws(w, i.ops[0], self.rs(w, i.ops[1]))
    def i_s_mov_b64(self, w, i): self.ws(w, i.ops[0], self.rs(w, i.ops[1], 2), 2)

    def _sarith(self, w, i, f, signed=False, carry=None):
        a, b = self.rs(w, i.ops[1]), self.rs(w, i.ops[2])
        if signed:
            a, b = _s32(a), _s32(b)
        r = f(a, b)
        self.ws(w, i.ops[0], r & 0xFFFFFFFF)
        return a, b, r

    def i_s_add_u32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a + b)
        w.scc = 1 if r > 0xFFFFFFFF else 0

    def i_s_addc_u32(self, w, i):
        c = w.scc
        a, b, r = self._sarith(w, i, lambda a, b: a + b + c)
        w.scc = 1 if r > 0xFFFFFFFF else 0

    def i_s_add_i32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a + b, signed=True)
        w.scc = 1 if not (-2 ** 31 <= r < 2 ** 31) else 0

    def i_s_sub_u32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a - b)
        w.scc = 1 if b > a else 0

    def i_s_sub_i32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a - b, signed=True)
        w.scc = 1 if not (-2 ** 31 <= r < 2 ** 31) else 0

    def i_s_mul_i32(self, w, i): self._sarith(w, i, lambda a, b: a * b, signed=True)
    def i_s_mul_hi_u32(self, w, i): self._sarith(w, i, lambda a, b: (a * b) >> 32)
    def i_s_mul_hi_i32(self, w, i): self._sarith(w, i, lambda a, b: (a * b) >> 32, signed=True)
    def i_s_cmp_le_u32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) <= self.rs(w, i.ops[1]))

    def i_s_lshl_b32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a << (b & 31))
        w.scc = 1 if (r & 0xFFFFFFFF) else 0

    def i_s_lshr_b32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a >> (b & 31))
        w.scc = 1 if (r & 0xFFFFFFFF) else 0

    def i_s_ashr_i32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a >> (b & 31), signed=True)
        w.scc = 1 if (r & 0xFFFFFFFF) else 0

    def i_s_and_b32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a & b)
        w.scc = 1 if r else 0

    def i_s_or_b32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a | b)
        w.scc = 1 if r else 0

    def i_s_xor_b32(self, w, i):
        a, b, r = self._sarith(w, i, lambda a, b: a ^ b)
        w.scc = 1 if r else 0

    def i_s_min_i32(self, w, i):
        a, b, r = self._sarith(w, i, min, signed=True)
        w.scc = 1 if a <= b else 0

    def i_s_max_i32(self, w, i):
        a, b, r = self._sarith(w, i, max, signed=True)
        w.scc = 1 if a >= b else 0

    def i_s_min_u32(self, w, i):
        a, b, r = self._sarith(w, i, min)
        w.scc = 1 if a <= b else 0

    def _s64(self, w, i, f):
        a, b = self.rs(w, i.ops[1], 2), self.rs(w, i.ops[2], 2)
        r = f(a, b) & 0xFFFFFFFFFFFFFFFF
        self.ws(w, i.ops[0], r, 2)
        w.scc = 1 if r else 0

    def i_s_and_b64(self, w, i): self._s64(w, i, lambda a, b: a & b)
    def i_s_or_b64(self, w, i): self._s64(w, i, lambda a, b: a | b)
    def i_s_andn2_b64(self, w, i): self._s64(w, i, lambda a, b: a & ~b)

    def i_s_cselect_b32(self, w, i):
        self.ws(w, i.ops[0], self.rs(w, i.ops[1]) if w.scc else self.rs(w, i.ops[2]))

    def i_s_cselect_b64(self, w, i):
        self.ws(w, i.ops[0], self.rs(w, i.ops[1], 2) if w.scc else self.rs(w, i.ops[2], 2), 2)

    def i_s_cmp_eq_u32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) == self.rs(w, i.ops[1]))
    def i_s_cmp_lg_u32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) != self.rs(w, i.ops[1]))
    def i_s_cmp_eq_i32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) == self.rs(w, i.ops[1]))
    def i_s_cmp_lg_i32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) != self.rs(w, i.ops[1]))
    def i_s_cmp_lt_i32(self, w, i): w.scc = int(_s32(self.rs(w, i.ops[0])) < _s32(self.rs(w, i.ops[1])))
    def i_s_cmp_le_i32(self, w, i): w.scc = int(_s32(self.rs(w, i.ops[0])) <= _s32(self.rs(w, i.ops[1])))
    def i_s_cmp_gt_i32(self, w, i): w.scc = int(_s32(self.rs(w, i.ops[0])) > _s32(self.rs(w, i.ops[1])))
    def i_s_cmp_ge_i32(self, w, i): w.scc = int(_s32(self.rs(w, i.ops[0])) >= _s32(self.rs(w, i.ops[1])))
    def i_s_cmp_lt_u32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) < self.rs(w, i.ops[1]))
    def i_s_cmp_ge_u32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) >= self.rs(w, i.ops[1]))
    def i_s_cmp_gt_u32(self, w, i): w.scc = int(self.rs(w, i.ops[0]) > self.rs(w, i.ops[1]))
    def i_s_cmp_lg_u64(self, w, i): w.scc = int(self.rs(w, i.ops[0], 2) != self.rs(w, i.ops[1], 2))
    def i_s_cmp_eq_u64(self, w, i): w.scc = int(self.rs(w, i.ops[0], 2) == self.rs(w, i.ops[1], 2))

    def _jump(self, w, i):
        lab = i.ops[0].val
        if lab not in self.labels:
            raise EmuError(f"unknown label {lab}")
        w.pc = self.labels[lab]

    def i_s_branch(self, w, i): self._jump(w, i)

    def i_s_cbranch_scc0(self, w, i):
        if not w.scc: self._jump(w, i)

    def i_s_cbranch_scc1(self, w, i):
        if w.scc: self._jump(w, i)

    def i_s_cbranch_vccz(self, w, i):
        if (int(w.s[106]) | int(w.s[107])) == 0: self._jump(w, i)

    def i_s_cbranch_vccnz(self, w, i):
        if (int(w.s[106]) | int(w.s[107])) != 0: self._jump(w, i)

    # ---- VALU -------------------------------------------------------------------------------------------------------
    def _valu(self, w, i, f, nsrc):
        srcs = [self.rv(w, i.ops[1 + k]) for k in range(nsrc)]
        r = np.asarray(f(*srcs)).astype(np.uint32)
        m = self.exec_mask(w)
        if m.all():
            self.wv(w, i.ops[0], r)
        else:
            old = self.rv(w, Op(i.ops[0].kind, i.ops[0].idx))
            self.wv(w, i.ops[0], np.where(m, r, old))

    def _fl(self, w, i, f, nsrc):
        with np.errstate(all="ignore"):
            self._valu(w, i, lambda *a: f2u(f(*[u2f(x) for x in a])), nsrc)

    def i_v_mov_b32(self, w, i): self._valu(w, i, lambda a: a, 1)
    def i_v_add_u32(self, w, i): self._valu(w, i, lambda a, b: a + b, 2)
    def i_v_sub_u32(self, w, i): self._valu(w, i, lambda a, b: a - b, 2)
    def i_v_subrev_u32(self, w, i): self._valu(w, i, lambda a, b: b - a, 2)
    def i_v_mul_lo_u32(self, w, i): self._valu(w, i, lambda a, b: (a.astype(np.uint64) * b.astype(np.uint64)) & 0xFFFFFFFF, 2)
    def i_v_mul_u32_u24(self, w, i): self._valu(w, i, lambda a, b: ((a & 0xFFFFFF).astype(np.uint64) * (b & 0xFFFFFF).astype(np.uint64)) & 0xFFFFFFFF, 2)
    def i_v_lshlrev_b32(self, w, i): self._valu(w, i, lambda a, b: b << (a & 31), 2)
    def i_v_lshrrev_b32(self, w, i): self._valu(w, i, lambda a, b: b >> (a & 31), 2)
    def i_v_and_b32(self, w, i): self._valu(w, i, lambda a, b: a & b, 2)
    def i_v_or_b32(self, w, i): self._valu(w, i, lambda a, b: a | b, 2)
    def i_v_xor_b32(self, w, i): self._valu(w, i, lambda a, b: a ^ b, 2)
    def i_v_lshl_add_u32(self, w, i): self._valu(w, i, lambda a, b, c: (a << (b & 31)) + c, 3)
    def i_v_lshl_or_b32(self, w, i): self._valu(w, i, lambda a, b, c: (a << (b & 31)) | c, 3)
    def i_v_add_lshl_u32(self, w, i): self._valu(w, i, lambda a, b, c: (a + b) << (c & 31), 3)
    def i_v_and_or_b32(self, w, i): self._valu(w, i, lambda a, b, c: (a & b) | c, 3)
    def i_v_add3_u32(self, w, i): self._valu(w, i, lambda a, b, c: a + b + c, 3)
    def i_v_mad_u32_u24(self, w, i): self._valu(w, i, lambda a, b, c: (((a & 0xFFFFFF).astype(np.uint64) * (b & 0xFFFFFF).astype(np.uint64)) + c) & 0xFFFFFFFF, 3)
    def i_v_min_u32(self, w, i): self._valu(w, i, np.minimum, 2)
    def i_v_max_u32(self, w, i): self._valu(w, i, np.maximum, 2)
    def i_v_min_i32(self, w, i): self._valu(w, i, lambda a, b: np.minimum(a.view(np.int32), b.view(np.int32)).view(np.uint32), 2)
    def i_v_max_i32(self, w, i): self._valu(w, i, lambda a, b: np.maximum(a.view(np.int32), b.view(np.int32)).view(np.uint32), 2)

    def i_v_add_f32(self, w, i): self._fl(w, i, lambda a, b: a + b, 2)
    def i_v_sub_f32(self, w, i): self._fl(w, i, lambda a, b: a - b, 2)
    def i_v_subrev_f32(self, w, i): self._fl(w, i, lambda a, b: b - a, 2)
    def i_v_mul_f32(self, w, i): self._fl(w, i, lambda a, b: a * b, 2)
    def i_v_fma_f32(self, w, i): self._fl(w, i, lambda a, b, c: (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32), 3)
    def i_v_max_f32(self, w, i): self._fl(w, i, np.fmax, 2)
    def i_v_min_f32(self, w, i): self._fl(w, i, np.fmin, 2)
    def i_v_max3_f32(self, w, i): self._fl(w, i, lambda a, b, c: np.fmax(np.fmax(a, b), c), 3)
    def i_v_exp_f32(self, w, i): self._fl(w, i, lambda a: np.exp2(a.astype(np.float64)).astype(np.float32), 1)
    def i_v_rcp_f32(self, w, i): self._fl(w, i, lambda a: (1.0 / a.astype(np.float64)).astype(np.float32), 1)
    def i_v_rsq_f32(self, w, i): self._fl(w, i, lambda a: (1.0 / np.sqrt(a.astype(np.float64))).astype(np.float32), 1)
    def i_v_log_f32(self, w, i): self._fl(w, i, lambda a: np.log2(a.astype(np.float64)).astype(np.float32), 1)

    def i_v_cvt_pk_bf16_f32(self, w, i):
        self._valu(w, i, lambda a, b: bf16_round(u2f(a)) | (bf16_round(u2f(b)) << 16), 2)

    def i_v_cvt_u32_f32(self, w, i):
        def f(a):
            x = np.nan_to_num(u2f(a).astype(np.float64), nan=0.0)
            return np.clip(np.trunc(x), 0, 4294967295).astype(np.uint32)
        self._valu(w, i, f, 1)

    def _pk(self, w, i, f):
        d, a, b = i.ops
        def part(op, k):
            return u2f(self.rv(w, Op(op.kind, op.idx + k, 1, op.val)) if op.kind != "imm" else self.rv(w, op))
        with np.errstate(all="ignore"):
            lo, hi = f(part(a, 0), part(b, 0)), f(part(a, 1), part(b, 1))
        self.wv(w, d, f2u(lo.astype(np.float32)), 0)
        self.wv(w, d, f2u(hi.astype(np.float32)), 1)

    def i_v_pk_mul_f32(self, w, i): self._pk(w, i, lambda a, b: a * b)
    def i_v_pk_add_f32(self, w, i): self._pk(w, i, lambda a, b: a + b)

    def i_v_pk_fma_f32(self, w, i):
        d, a, b, c = i.ops
        def part(op, k):
            return u2f(self.rv(w, Op(op.kind, op.idx + k, 1, op.val)) if op.kind != "imm" else self.rv(w, op)).astype(np.float64)
        with np.errstate(all="ignore"):
            lo, hi = part(a, 0) * part(b, 0) + part(c, 0), part(a, 1) * part(b, 1) + part(c, 1)      # fused: one rounding
        self.wv(w, d, f2u(lo.astype(np.float32)), 0)
        self.wv(w, d, f2u(hi.astype(np.float32)), 1)

    def i_v_dot2c_f32_bf16(self, w, i):
        d, a, b = i.ops
        A, B, D = self.rv(w, a), self.rv(w, b), u2f(self.rv(w, d)).astype(np.float64)
        lo = bf16_to_f32(A & U32(0xFFFF)).astype(np.float64) * bf16_to_f32(B & U32(0xFFFF)).astype(np.float64)
        hi = bf16_to_f32(A >> U32(16)).astype(np.float64) * bf16_to_f32(B >> U32(16)).astype(np.float64)
        with np.errstate(all="ignore"):
            self.wv(w, d, f2u((lo + hi + D).astype(np.float32)))

    def i_v_cvt_f32_u32(self, w, i): self._valu(w, i, lambda a: f2u(a.astype(np.float32)), 1)
    def i_v_cvt_f32_i32(self, w, i): self._valu(w, i, lambda a: f2u(a.view(np.int32).astype(np.float32)), 1)

    def i_v_lshl_add_u64(self, w, i):
        d, a, sh, b = i.ops
        lo = lambda op, k: self.rv(w, Op(op.kind, op.idx + k, 1, op.val)) if op.kind != "imm" else np.full(64, (op.val >> (32 * k)) & 0xFFFFFFFF if k == 0 else 0, dtype=np.uint32)
        A = lo(a, 0).astype(np.uint64) | (lo(a, 1).astype(np.uint64) << 32)
        B = lo(b, 0).astype(np.uint64) | (lo(b, 1).astype(np.uint64) << 32)
        s = int(self.rv(w, sh)[0]) & 7 if sh.kind != "imm" else sh.val & 7
        r = (A << np.uint64(s)) + B
        self.wv(w, d, (r & np.uint64(0xFFFFFFFF)).astype(np.uint32), 0)
        self.wv(w, d, (r >> np.uint64(32)).astype(np.uint32), 1)

    def _cmp(self, w, i, f, kind):
        dst = i.ops[0] if len(i.ops) == 3 else Op("s", 106, 2)
        a, b = (i.ops[1], i.ops[2]) if len(i.ops) == 3 else (i.ops[0], i.ops[1])
        x, y = self.rv(w, a), self.rv(w, b)
        if kind == "f":
            x, y = u2f(x), u2f(y)
        elif kind == "i":
            x, y = x.view(np.int32), y.view(np.int32)
        with np.errstate(all="ignore"):
            r = f(x, y) & self.exec_mask(w)
        v = 0
        for l in range(64):
            if r[l]:
                v |= 1 << l
        self.ws(w, dst, v, 2)

    def i_v_cmp_gt_f32(self, w, i): self._cmp(w, i, lambda a, b: a > b, "f")
    def i_v_cmp_lt_f32(self, w, i): self._cmp(w, i, lambda a, b: a < b, "f")
    def i_v_cmp_ge_f32(self, w, i): self._cmp(w, i, lambda a, b: a >= b, "f")
    def i_v_cmp_neq_f32(self, w, i): self._cmp(w, i, lambda a, b: ~(a == b), "f")
    def i_v_cmp_gt_i32(self, w, i): self._cmp(w, i, lambda a, b: a > b, "i")
    def i_v_cmp_lt_i32(self, w, i): self._cmp(w, i, lambda a, b: a < b, "i")
    def i_v_cmp_ge_i32(self, w, i): self._cmp(w, i, lambda a, b: a >= b, "i")
    def i_v_cmp_le_i32(self, w, i): self._cmp(w, i, lambda a, b: a <= b, "i")
    def i_v_cmp_gt_u32(self, w, i): self._cmp(w, i, lambda a, b: a > b, "u")
    def i_v_cmp_lt_u32(self, w, i): self._cmp(w, i, lambda a, b: a < b, "u")
    def i_v_cmp_ge_u32(self, w, i): self._cmp(w, i, lambda a, b: a >= b, "u")
    def i_v_cmp_le_u32(self, w, i): self._cmp(w, i, lambda a, b: a <= b, "u")

    def i_v_cmp_eq_u32(self, w, i): self._cmp(w, i, lambda a, b: a == b, "u")
    def i_v_cmp_ne_u32(self, w, i): self._cmp(w, i, lambda a, b: a != b, "u")

    def i_v_cndmask_b32(self, w, i):
        sel = i.ops[3] if len(i.ops) > 3 else Op("s", 106, 2)
        m = self.rs(w, sel, 2)
        bits = np.array([(m >> l) & 1 for l in range(64)], dtype=bool)
        self._valu(w, i, lambda a, b: np.where(bits, b, a), 2)

    def i_v_accvgpr_read_b32(self, w, i): self._valu(w, i, lambda a: a, 1)
    def i_v_accvgpr_write_b32(self, w, i): self._valu(w, i, lambda a: a, 1)
    def i_v_accvgpr_read(self, w, i): self._valu(w, i, lambda a: a, 1)
    def i_v_accvgpr_write(self, w, i): self._valu(w, i, lambda a: a, 1)

    def i_v_readfirstlane_b32(self, w, i):
        self.ws(w, i.ops[0], int(self.rv(w, i.ops[1])[0]))

    def i_v_mbcnt_lo_u32_b32(self, w, i):
        m = self.rv(w, i.ops[1]); c = self.rv(w, i.ops[2])
        lanes = np.arange(64)
        r = np.array([bin(int(m[l]) & ((1 << min(l, 32)) - 1)).count("1") for l in lanes], dtype=np.uint32) + c
        self.wv(w, i.ops[0], r)

    def i_v_mbcnt_hi_u32_b32(self, w, i):
        m = self.rv(w, i.ops[1]); c = self.rv(w, i.ops[2])
        r = np.array([bin(int(m[l]) & ((1 << max(0, l - 32)) - 1)).count("1") for l in range(64)], dtype=np.uint32) + c
        self.wv(w, i.ops[0], r)

    def i_v_permlane32_swap_b32(self, w, i):
        self.full_exec(w, "v_permlane32_swap")
        d, s = self.rv(w, i.ops[0]).copy(), self.rv(w, i.ops[1]).copy()
        d2, s2 = d.copy(), s.copy()
        d2[32:], s2[:32] = s[:32], d[32:]
        self.wv(w, i.ops[0], d2); self.wv(w, i.ops[1], s2)

    def i_v_permlane16_swap_b32(self, w, i):
        self.full_exec(w, "v_permlane16_swap")
        d, s = self.rv(w, i.ops[0]).copy(), self.rv(w, i.ops[1]).copy()
        d2, s2 = d.copy(), s.copy()
        d2[16:32], s2[0:16] = s[0:16], d[16:32]
        d2[48:64], s2[32:48] = s[32:48], d[48:64]
        self.wv(w, i.ops[0], d2); self.wv(w, i.ops[1], s2)

    # ---- MFMA -------------------------------------------------------------------------------------------------------
    def _frag_bf16(self, w, op):
        """4 dwords x 64 lanes -> float32 [64 lanes][8 elements]"""
        out = np.zeros((64, 8), dtype=np.float32)
        for d in range(4):
            x = self.rv(w, op, d)
            out[:, 2 * d] = bf16_to_f32(x & U32(0xFFFF))
            out[:, 2 * d + 1] = bf16_to_f32(x >> 16)
        return out

    def i_v_mfma_f32_32x32x16_bf16(self, w, i):
        self.full_exec(w, "mfma")
        D, A, B, C = i.ops
        fa, fb = self._frag_bf16(w, A), self._frag_bf16(w, B)
        lanes = np.arange(64)
        Am = np.zeros((32, 16), dtype=np.float64); Bm = np.zeros((16, 32), dtype=np.float64)
        for j in range(8):
            Am[lanes & 31, 8 * (lanes >> 5) + j] = fa[:, j]
            Bm[8 * (lanes >> 5) + j, lanes & 31] = fb[:, j]
        with np.errstate(all="ignore"):
            P = Am @ Bm
        res = []
        for r in range(16):
            rows = (r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5)
            c = np.zeros(64, dtype=np.float32) if C.kind == "imm" else u2f(self.rv(w, C, r))
            with np.errstate(all="ignore"):
                res.append(f2u((c.astype(np.float64) + P[rows, lanes & 31]).astype(np.float32)))
        for r in range(16):
            self.wv(w, D, res[r], r)

    def i_v_mfma_i32_32x32x32_i8(self, w, i):
        """A / B: 16 int8 per lane (4 dwords): A[row l & 31][k = 16 (l >> 5) + j], B[k][col l & 31]; int32 accumulate; C/D layout as
        every 32x32 form.  (The k order inside the instruction is not documented for i8; any order common to A and B gives the
        same sums, and the GEMM feeds both operands through the same addressing.)"""
        self.full_exec(w, "mfma")
        D, A, B, C = i.ops
        lanes = np.arange(64)
        Am = np.zeros((32, 32), dtype=np.int64); Bm = np.zeros((32, 32), dtype=np.int64)
        for d in range(4):
            xa, xb = self.rv(w, A, d), self.rv(w, B, d)
            for b in range(4):
                k = 16 * (lanes >> 5) + 4 * d + b
                Am[lanes & 31, k] = ((xa >> (8 * b)) & U32(0xFF)).astype(np.uint8).view(np.int8)
                Bm[k, lanes & 31] = ((xb >> (8 * b)) & U32(0xFF)).astype(np.uint8).view(np.int8)
        P = Am @ Bm
        res = []
        for r in range(16):
            rows = (r & 3) + 8 * (r >> 2) + 4 * (lanes >> 5)
            c = np.zeros(64, dtype=np.int64) if C.kind == "imm" else self.rv(w, C, r).view(np.int32).astype(np.int64)
            res.append(((c + P[rows, lanes & 31]) & 0xFFFFFFFF).astype(np.uint32))
        for r in range(16):
            self.wv(w, D, res[r], r)

    def i_v_mfma_f32_16x16x32_bf16(self, w, i):
        self.full_exec(w, "mfma")
        D, A, B, C = i.ops
        fa, fb = self._frag_bf16(w, A), self._frag_bf16(w, B)
        lanes = np.arange(64)
        Am = np.zeros((16, 32), dtype=np.float64); Bm = np.zeros((32, 16), dtype=np.float64)
        for j in range(8):
            Am[lanes & 15, 8 * (lanes >> 4) + j] = fa[:, j]
            Bm[8 * (lanes >> 4) + j, lanes & 15] = fb[:, j]
        with np.errstate(all="ignore"):
            P = Am @ Bm
        res = []
        for r in range(4):
            rows = 4 * (lanes >> 4) + r
            c = np.zeros(64, dtype=np.float32) if C.kind == "imm" else u2f(self.rv(w, C, r))
            with np.errstate(all="ignore"):
                res.append(f2u((c.astype(np.float64) + P[rows, lanes & 15]).astype(np.float32)))
        for r in range(4):
            self.wv(w, D, res[r], r)

    # ---- LDS --------------------------------------------------------------------------------------------------------
    def _lds_addr(self, w, i, op):
        a = self.rv(w, op).astype(np.int64) + int(i.mods.get("offset", 0))
        if (a < 0).any() or (a >= self.lds.size).any():
            raise EmuError(f"LDS address out of range: {a.min()}..{a.max()}")
        return a

    def _ds_read(self, w, i, nbytes):
        addr = self._lds_addr(w, i, i.ops[1])
        if (addr % min(nbytes, 16) != 0).any() and nbytes >= 8:
            if (addr % 8 != 0).any():
                raise EmuError("unaligned LDS read")
        dst = i.ops[0]
        nd = nbytes // 4
        for d in range(nd):
            self.wv(w, dst, np.full(64, POISON, dtype=np.uint32), d)

        def complete():
            idx = addr[:, None] + np.arange(nbytes)[None, :]
            words = np.ascontiguousarray(self.lds[idx]).view(np.uint32)          # [64][nd]
            for d in range(nd):
                self.wv(w, dst, words[:, d], d)
        self._issue(w, "lgkm", complete)

    def i_ds_read_b128(self, w, i): self._ds_read(w, i, 16)
    def i_ds_read_b64(self, w, i): self._ds_read(w, i, 8)
    def i_ds_read_b32(self, w, i): self._ds_read(w, i, 4)

    def i_ds_read_b64_tr_b16(self, w, i):
        self.full_exec(w, "ds_read_b64_tr_b16")
        addr = self._lds_addr(w, i, i.ops[1])
        if (addr % 8 != 0).any():
            raise EmuError("ds_read_b64_tr_b16 address not 8-byte aligned")
        dst = i.ops[0]
        for d in range(2):
            self.wv(w, dst, np.full(64, POISON, dtype=np.uint32), d)

        def complete():
            lanes = np.arange(64)
            g, li = lanes >> 4, lanes & 15
            out = np.zeros((64, 4), dtype=np.uint32)          # 4 x 16-bit elements per lane
            for q in range(4):
                src = 16 * g + 4 * q + (li >> 2)
                o = addr[src] + 2 * (li & 3)
                out[:, q] = self.lds[o].astype(np.uint32) | (self.lds[o + 1].astype(np.uint32) << 8)
            self.wv(w, dst, out[:, 0] | (out[:, 1] << 16), 0)
            self.wv(w, dst, out[:, 2] | (out[:, 3] << 16), 1)
        self._issue(w, "lgkm", complete)

    def _ds_write(self, w, i, nbytes):
        addr = self._lds_addr(w, i, i.ops[0])
        nd = nbytes // 4
        data = [self.rv(w, i.ops[1], d).copy() for d in range(nd)]
        m = self.exec_mask(w)

        def complete():
            for l in range(64):
                if m[l]:
                    for d in range(nd):
                        o = int(addr[l]) + 4 * d
                        self.lds[o:o + 4] = np.frombuffer(int(data[d][l]).to_bytes(4, "little"), dtype=np.uint8)
        self._issue(w, "lgkm", complete)

    def i_ds_write_b128(self, w, i): self._ds_write(w, i, 16)
    def i_ds_write_b64(self, w, i): self._ds_write(w, i, 8)
    def i_ds_write_b32(self, w, i): self._ds_write(w, i, 4)

    # ---- VMEM -------------------------------------------------------------------------------------------------------
    def _gaddr(self, w, i, vop, sop):
        """global_* addressing: 64-bit vaddr (+ offset) when saddr is `off`, else saddr (64-bit SGPR) + 32-bit voffset"""
        off = int(i.mods.get("offset", 0))
        if off >= 4096:
            off -= 8192
        if sop.kind == "off":
            lo, hi = self.rv(w, Op("v", vop.idx)), self.rv(w, Op("v", vop.idx + 1))
            return (lo.astype(np.uint64) | (hi.astype(np.uint64) << np.uint64(32))).astype(np.int64) + off
        base = self.rs(w, sop, 2)
        return self.rv(w, Op("v", vop.idx)).astype(np.int64) + base + off

    def _vload(self, w, i, dst, addrs, nbytes, valid=None):
        nd = nbytes // 4
        m = self.exec_mask(w)
        for d in range(nd):
            self.wv(w, dst, np.where(m, POISON, self.rv(w, Op(dst.kind, dst.idx + d))), d)

        def complete():
            for d in range(nd):
                vals = self.rv(w, Op(dst.kind, dst.idx + d)).copy()
                for l in range(64):
                    if not m[l]:
                        continue
                    if valid is not None and not valid[l]:
                        vals[l] = 0
                    else:
                        vals[l] = int.from_bytes(self.mem.read(int(addrs[l]) + 4 * d, 4).tobytes(), "little")
                self.wv(w, dst, vals, d)
        self._issue(w, "vm", complete)

    def i_global_load_dwordx4(self, w, i): self._vload(w, i, i.ops[0], self._gaddr(w, i, i.ops[1], i.ops[2]), 16)
    def i_global_load_dwordx2(self, w, i): self._vload(w, i, i.ops[0], self._gaddr(w, i, i.ops[1], i.ops[2]), 8)
    def i_global_load_dword(self, w, i): self._vload(w, i, i.ops[0], self._gaddr(w, i, i.ops[1], i.ops[2]), 4)

    def _vstore(self, w, i, addrs, src, nbytes):
        nd = nbytes // 4
        data = [self.rv(w, src, d).copy() for d in range(nd)]
        m = self.exec_mask(w)

        def complete():
            for l in range(64):
                if m[l]:
                    for d in range(nd):
                        self.mem.write(int(addrs[l]) + 4 * d, np.frombuffer(int(data[d][l]).to_bytes(4, "little"), dtype=np.uint8))
        self._issue(w, "vm", complete)

    def i_global_store_dwordx4(self, w, i): self._vstore(w, i, self._gaddr(w, i, i.ops[0], i.ops[2]), i.ops[1], 16)
    def i_global_store_dwordx2(self, w, i): self._vstore(w, i, self._gaddr(w, i, i.ops[0], i.ops[2]), i.ops[1], 8)
    def i_global_store_dword(self, w, i): self._vstore(w, i, self._gaddr(w, i, i.ops[0], i.ops[2]), i.ops[1], 4)

    def _lds_dma(self, w, i, addrs, valid):
        """16 bytes per lane -> LDS at M0[15:0]?? (full M0 used) + instruction offset + 16 * lane"""
        self.full_exec(w, "LDS-DMA")
        base = int(w.s[124]) + int(i.mods.get("offset", 0))      # the instruction offset moves the LDS destination too (as the source)
        if base + 1024 > self.lds.size:
            raise EmuError(f"LDS-DMA destination {base:#x} out of range")

        def complete():
            for l in range(64):
                if valid is not None and not valid[l]:
                    self.lds[base + 16 * l: base + 16 * l + 16] = 0
                else:
                    self.lds[base + 16 * l: base + 16 * l + 16] = self.mem.read(int(addrs[l]), 16)
        self._issue(w, "vm", complete)

    def i_global_load_lds_dwordx4(self, w, i):
        self._lds_dma(w, i, self._gaddr(w, i, i.ops[0], i.ops[1]), None)

    def i_buffer_load_dwordx4(self, w, i):
        """buffer_load_dwordx4 vdst, voffset, s[rsrc:rsrc+3], soffset offen [offset:imm] [lds]   (raw buffer: stride 0).
        With `lds` there is no vdst: operands are voffset, srsrc, soffset.  Range check (gfx9 raw buffer): voffset + imm
        against num_records -- soffset is NOT part of the check."""
        lds = bool(i.mods.get("lds"))
        ops = i.ops
        if lds:
            voff, rsrc, soff = ops[0], ops[1], ops[2]
        else:
            dst, voff, rsrc, soff = ops
        if not i.mods.get("offen"):
            raise EmuError("only the offen form is modelled")
        base = int(w.s[rsrc.idx]) | ((int(w.s[rsrc.idx + 1]) & 0xFFFF) << 32)
        nrec = int(w.s[rsrc.idx + 2])
        imm = int(i.mods.get("offset", 0))
        vo = self.rv(w, voff).astype(np.int64) + imm
        valid = (vo + 16) <= nrec
        addrs = base + vo + self.rs(w, soff)
        if lds:
            self._lds_dma(w, i, addrs, valid)
        else:
            self._vload(w, i, dst, addrs, 16, valid)


def _s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x
