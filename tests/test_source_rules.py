"""Static rules over the HIP sources (no GPU, no compiler).

Rule 1 (experiments/README.md, round 2; profiles/r02_abort_splitk.md): hipcc does not count the memory operations of an
`asm volatile` statement, so a load with a VGPR destination whose wait is not inside the SAME statement can deliver its data
into a register the allocator has already given to something else (a GPU memory fault when that something was an address).
Every asm statement in the library that contains a register-destination `global_load` / `buffer_load` / `flat_load` /
`scratch_load` must therefore carry its own `s_waitcnt vmcnt(...)`.  LDS-DMA forms (`global_load_lds_*`, `buffer_load ... lds`)
have no register destination and are exempt (their completion is counted by hand with vmcnt before a barrier)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = sorted(glob.glob(os.path.join(ROOT, "longlive_amd", "csrc", "*.hip")) +
                 glob.glob(os.path.join(ROOT, "longlive_amd", "csrc", "*.h")))


def asm_statements(text):
    """Yields (line_no, statement text) of every `asm volatile( ... );` / `asm( ... );` statement."""
    for m in re.finditer(r"\basm\s*(?:volatile)?\s*\(", text):
        depth, i = 1, m.end()
        in_str = False
        while i < len(text) and depth:
            c = text[i]
            if in_str:
                if c == "\\":
                    i += 1
                elif c == '"':
                    in_str = False
            elif c == '"':
                in_str = True
            elif c == "(":
                depth += 1
            elif c == ")":
                depth -= 1
            i += 1
        yield text.count("\n", 0, m.start()) + 1, text[m.start():i]


def strings_of(stmt):
    """The assembler template of one asm statement: its string literals up to the first ':' outside a string, joined with
    newlines (adjacent literals are separate instructions more often than not; a word must never fuse across them)."""
    parts, cur, in_str, i = [], [], False, stmt.index("(") + 1
    while i < len(stmt):
        c = stmt[i]
        if in_str:
            if c == "\\":
                cur.append(stmt[i:i + 2])
                i += 1
            elif c == '"':
                in_str = False
                parts.append("".join(cur))
                cur = []
            else:
                cur.append(c)
        elif c == '"':
            in_str = True
        elif c == ":":
            break
        i += 1
    return "\n".join(parts)


LOAD = re.compile(r"\b(global_load|buffer_load|flat_load|scratch_load)_(?!lds)\w+\b([^\\\n\"]*)")


def violations(text):
    out = []
    for line, stmt in asm_statements(text):
        body = strings_of(stmt)
        reg_loads = [m for m in LOAD.finditer(body) if not re.search(r"\blds\b", m.group(2))]
        if reg_loads and not re.search(r"s_waitcnt[^\\\n]*vmcnt\(", body):
            out.append((line, reg_loads[0].group(0).strip()))
    return out


def test_the_scanner_sees_what_it_should():
    bad = 'asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(a) : "v"(p) : "memory");'
    good = 'asm volatile("global_load_dword %0, %1, off sc1\\n\\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(pf) : "memory");'
    dma = 'asm volatile("s_mov_b32 m0, %1\\n\\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(l) : "memory");'
    dma2 = 'asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(o), "s"(r) : "memory");'
    assert len(violations(bad)) == 1
    assert violations(good) == [] and violations(dma) == [] and violations(dma2) == []


def test_every_asm_register_load_waits_in_its_own_statement():
    assert SOURCES, "no sources found"
    found = {}
    for path in SOURCES:
        v = violations(open(path).read())
        if v:
            found[os.path.relpath(path, ROOT)] = v
    assert not found, f"asm register loads without s_waitcnt vmcnt in the same statement: {found}"
