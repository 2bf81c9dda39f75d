#!/usr/bin/env python3
"""Per-kernel summary of the in-pipeline counter passes of tools/pmc_inpipe.sh (rocprofv3 --pmc over bench.py).

Only the STEADY-STATE part of each run is used: bench.py runs 4 warm-up + 2 timed blocks, every block issues the same launch
sequence, so the last third of the dispatches are the two steady-state blocks (Lk = 18720, roll + insert).  Kernels are grouped
by (name, grid size); per group the mean over its dispatches of

  duration            kernel-trace pass (un-counted run), End - Start
  cycles              GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the counter over the 8 XCDs)                 -- of the COUNTED run
  implied clock       cycles / duration of the same counted dispatches (the counter csv carries their timestamps): reads HIGH on
                      launches shorter than ~0.3 ms (MI355X_MICROARCH.md, DVFS give-back); counted passes serialise dispatches and
                      run ~3 % lower clocks than un-profiled ones
  MFMA utilisation    SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles); busy cycles per MFMA printed as a check (32 / 16)
  wave-cycle shares   SQ_WAIT_ANY (parked at a wait/barrier), SQ_WAIT_INST_ANY (issue-stalled), SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  fabric bytes        2 x FETCH_SIZE + WRITE_SIZE from TCC_EA0_RDREQ / _32B / TCC_BUBBLE and TCC_EA0_WRREQ / _64B (gfx950: FETCH_SIZE
                      reports half of a wide coalesced read stream; L2-miss bytes, Infinity-Cache hits included)

    python tools/pmc_inpipe_summary.py <outdir> [--md out.md] [--json out.json]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name


def label(name, us):
    """one kernel name, two launch shapes per layer: told apart by the dispatch's own duration"""
    if us is None:
        return name
    if "flash_attn_asm_kernel" in name:
        return name + (" [self, Lk=18720]" if us > 100 else " [cross, Lk=512]")
    if "gemm_asm_128_gate_res" in name:
        return name + (" [FFN2, K=8960]" if us > 60 else " [O, K=1536]")
    return name


def load_counters(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    if not files:
        return None
    per = defaultdict(dict)
    meta = {}
    for r in csv.DictReader(open(files[0])):
        i = int(r["Dispatch_Id"])
        meta[i] = short(r["Kernel_Name"])
        per[i][r["Counter_Name"]] = per[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("End_Timestamp") and r.get("Start_Timestamp"):       # the counted run's own duration of this dispatch
            per[i]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for i in meta:
        meta[i] = label(meta[i], per[i].get("_us"))
    return per, meta


def steady(ids):
    ids = sorted(ids)
    return set(ids[len(ids) - len(ids) // 3:])


def group_means(per, meta):
    keep = steady(per.keys())
    acc = defaultdict(lambda: defaultdict(list))
    for i in keep:
        for k, v in per[i].items():
            acc[meta[i]][k].append(v)
    return {g: {k: sum(v) / len(v) for k, v in c.items()} | {"_n": len(next(iter(c.values())))} for g, c in acc.items()}


def load_trace(d):
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        return {}
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) - len(rows) // 3:]
    acc = defaultdict(list)
    for r in rows:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        acc[label(short(r["Kernel_Name"]), us)].append(us)
    span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) * 1e-6
    return {g: (sum(v) / len(v), len(v)) for g, v in acc.items()}, span


def main():
    argv = sys.argv[1:]
    opts = {}
    for flag in ("--md", "--json"):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    out = argv[0]
    tr = load_trace(os.path.join(out, "trace"))
    trace, span = tr if tr else ({}, None)
    tables = {}
    for name in ("sq1", "sq2", "tcc_rd", "tcc_wr"):
        lc = load_counters(os.path.join(out, name))
        if lc:
            tables[name] = group_means(*lc)
    if "sq2" in tables:                               # wave-cycle shares come from their own pass: merge per kernel
        for g_, c_ in tables["sq2"].items():
            tables.setdefault("sq1", {}).setdefault(g_, {}).update({k: v for k, v in c_.items() if k not in ("GRBM_GUI_ACTIVE", "_n") or k not in tables["sq1"].get(g_, {})})
    groups = sorted(trace, key=lambda g: -trace[g][0] * trace[g][1]) if trace else sorted(tables.get("sq1", {}))
    total_us = sum(trace[g][0] * trace[g][1] for g in trace) if trace else None
    rec = {"steady_window_ms": span, "kernels": []}
    lines = ["| kernel | launches | avg us (trace) | share | MFMA util | cyc/MFMA | implied clock GHz | wait / stall / active | fabric MB per launch |",
             "|---|---|---|---|---|---|---|---|---|"]
    for g in groups:
        us, n = trace.get(g, (None, 0))
        d = {"kernel": g, "launches_in_window": n, "avg_us": us}
        sq = tables.get("sq1", {}).get(g)
        if sq and sq.get("GRBM_GUI_ACTIVE"):
            cyc = sq["GRBM_GUI_ACTIVE"] / 8
            d["cycles_counted"] = cyc
            if sq.get("_us"):
                d["avg_us_counted"] = sq["_us"]
                d["implied_clock_ghz"] = cyc / (sq["_us"] * 1e3)         # cycles and duration of the SAME (counted) dispatches
            if sq.get("SQ_INSTS_MFMA"):
                d["mfma_util"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
                d["busy_cycles_per_mfma"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"] / sq["SQ_INSTS_MFMA"]
            if sq.get("SQ_WAVE_CYCLES"):
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    d[k + "_share"] = sq.get(k, 0.0) / sq["SQ_WAVE_CYCLES"]
        rd, wr = tables.get("tcc_rd", {}).get(g), tables.get("tcc_wr", {}).get(g)
        if rd and wr:
            gg = lambda c, n_: c.get(n_, c.get(n_ + "_sum", 0.0))
            bub, rd32 = gg(rd, "TCC_BUBBLE"), gg(rd, "TCC_EA0_RDREQ_32B")
            fetch = bub * 128 + (gg(rd, "TCC_EA0_RDREQ") - bub - rd32) * 64 + rd32 * 32
            w64 = gg(wr, "TCC_EA0_WRREQ_64B")
            write = (gg(wr, "TCC_EA0_WRREQ") - w64) * 32 + w64 * 64
            d["fabric_bytes_per_launch"] = 2 * fetch + write
        rec["kernels"].append(d)
        f = lambda x, fmt: "-" if x is None else fmt % x
        share = None if not (us and total_us) else us * n / total_us
        lines.append("| `%s` | %d | %s | %s | %s | %s | %s | %s | %s |" % (
            g[:96], n, f(us, "%.1f"), f(share and 100 * share, "%.1f %%"), f(d.get("mfma_util") and 100 * d["mfma_util"], "%.1f %%"),
            f(d.get("busy_cycles_per_mfma"), "%.1f"), f(d.get("implied_clock_ghz"), "%.2f"),
            "-" if "SQ_WAIT_ANY_share" not in d else "%.0f / %.0f / %.0f %%" % (100 * d["SQ_WAIT_ANY_share"], 100 * d["SQ_WAIT_INST_ANY_share"], 100 * d["SQ_ACTIVE_INST_ANY_share"]),
            f(d.get("fabric_bytes_per_launch") and d["fabric_bytes_per_launch"] / 1e6, "%.1f")))
    text = "\n".join(lines)
    print(text)
    if "--md" in opts:
        open(opts["--md"], "w").write(text + "\n")
    if "--json" in opts:
        json.dump(rec, open(opts["--json"], "w"), indent=1)


if __name__ == "__main__":
    main()
