#!/usr/bin/env python3
"""Joins rocprofv3 --pmc passes (counter_collection.csv) over `tools/kbench shipped` with kbench's own "#WORK" records
(algorithmic FLOPs / bytes and un-profiled time per launch) into the per-kernel evidence table of DESIGN.md:

  MFMA utilisation  = SQ_VALU_MFMA_BUSY_CYCLES / f / (SIMDs x kernel cycles)
        kernel cycles = GRBM_GUI_ACTIVE / 8         (rocprofv3 sums the counter over the 8 XCDs)
        SIMDs         = 256 CUs x 4
        f             = sampling factor of the SQ block in this collection mode = SQ_INSTS_MFMA counted / MFMAs the launch
                        needs (algorithmic FLOPs / FLOPs per MFMA).  rocprofv3 7.2 on gfx950 reports SQ counters from a
                        subset of the shader engines, so f < 1; it is printed, not assumed.
        (SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = cycles the matrix pipe is busy per MFMA: 32 for 32x32x16 bf16,
         16 for 16x16x32 bf16 / 16x16x64 i8 -- also printed as a consistency check.)
  HBM bytes / launch = 2 x FETCH_SIZE + WRITE_SIZE with
        FETCH_SIZE = TCC_BUBBLE*128 + (TCC_EA0_RDREQ - TCC_BUBBLE - TCC_EA0_RDREQ_32B)*64 + TCC_EA0_RDREQ_32B*32
        WRITE_SIZE = (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)*32 + TCC_EA0_WRREQ_64B*64
        (MI355X_MICROARCH.md: gfx950's FETCH_SIZE reports exactly half of a wide coalesced read stream; WRITE_SIZE is exact;
         the derived FETCH_SIZE / WRITE_SIZE counters themselves crash rocprofv3 7.2 on this pool, so the base counters are
         collected in separate passes and the published expressions are evaluated here).  These are L2-miss (fabric) bytes:
         Infinity-Cache hits are included, so they bound HBM traffic from above.
  HBM GB/s          = bytes per launch / un-profiled launch time, as a fraction of 8 TB/s.

Dispatches are matched to kbench's launches IN ORDER: kbench runs its entries one after the other (3 warm-up + N timed
launches each), so consecutive dispatches with the same kernel name form one entry even when two entries share a kernel
(O-projection and FFN2 both run gemm_kernel_v2<2, false> on 228 workgroups).

    python tools/pmc_summary.py --work kbench.log dir_or_csv [dir_or_csv ...] [--json out.json] [--md out.md]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

SIMDS = 1024
XCDS = 8
HBM_PEAK = 8.0e12
FLOP_PER_MFMA_CYCLE = {"bf16": 1024.0, "i8": 2048.0}      # per SIMD per cycle (32768 / 32, 16384 / 16; int8 twice that)


def runs_of(path):
    """[(kernel_name, {counter: mean over the run's dispatches})] in dispatch order for ONE csv (= one pmc pass)."""
    rows = defaultdict(dict)
    names = {}
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatch_Id"])
        names[d] = r["Kernel_Name"]
        rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    out = []
    for d in sorted(rows):
        if out and out[-1][0] == names[d]:
            out[-1][1].append(rows[d])
        else:
            out.append((names[d], [rows[d]]))
    res = []
    for name, lst in out:
        keys = set().union(*[set(x) for x in lst])
        use = lst[3:] if len(lst) > 4 else lst              # drop kbench's 3 warm-up launches
        res.append((name, {k: sum(x.get(k, 0.0) for x in use) / len(use) for k in keys}, len(lst)))
    return res


def main():
    argv = sys.argv[1:]
    opts = {}
    for flag in ("--work", "--json", "--md"):
        if flag in argv:
            i = argv.index(flag)
            opts[flag] = argv[i + 1]
            del argv[i:i + 2]
    files = []
    for a in argv:
        files += sorted(glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)) if os.path.isdir(a) else [a]
    work = []
    if "--work" in opts:
        for line in open(opts["--work"]):
            if line.startswith("#WORK "):
                work.append(json.loads(line[6:]))
    table = {w["tag"]: dict(w) for w in work}
    for f in files:
        runs = runs_of(f)
        pos = 0
        for w in work:
            for j in range(pos, len(runs)):
                if w["match"] in runs[j][0]:
                    t = table[w["tag"]]
                    t.setdefault("kernel", re.sub(r"\(.*", "", runs[j][0]).replace("void ", ""))
                    t.setdefault("counters", {}).update(runs[j][1])
                    t["dispatches_seen"] = runs[j][2]
                    pos = j + 1
                    break
    out = {}
    for tag, t in table.items():
        c = t.get("counters", {})
        g = lambda n: c.get(n, c.get(n + "_sum"))
        d = {"kernel": t.get("kernel", t["match"]), "bound": t["bound"], "us_unprofiled": t["us"], "flop": t["flop"],
             "algorithmic_bytes": t["bytes"]}
        if g("GRBM_GUI_ACTIVE"):
            cyc = g("GRBM_GUI_ACTIVE") / XCDS
            d["kernel_cycles_profiled"] = cyc
        if t["bound"] == "mfma" and g("SQ_INSTS_MFMA") and g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
            busy_per = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_INSTS_MFMA")
            flop_per_mfma = 32768.0 if abs(busy_per - 32) < 4 else 16384.0        # 32x32x16 vs 16x16x32 (bf16)
            need = t["flop"] / flop_per_mfma
            f = g("SQ_INSTS_MFMA") / need
            d.update(mfma_busy_cycles_per_mfma=busy_per, sq_sampling_factor=f,
                     mfma_util=g("SQ_VALU_MFMA_BUSY_CYCLES") / f / (SIMDS * d["kernel_cycles_profiled"]),
                     clock_ghz_profiled=None)
            for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES",
                      "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS",
                      "SQ_VALU_MFMA_COEXEC_CYCLES"):
                if g(k) is not None:
                    d[k] = g(k)
            if g("SQ_WAVE_CYCLES"):
                for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if g(k) is not None:
                        d[k + "_share"] = g(k) / g("SQ_WAVE_CYCLES")
            # time-based figure from the UN-profiled launch (profiled passes run at a lower clock)
            d["achieved_tflops"] = t["flop"] / (t["us"] * 1e-6) / 1e12
            d["frac_of_2.5PF"] = d["achieved_tflops"] / 2500.0
        if g("TCC_EA0_RDREQ") is not None:
            bub, rd32 = g("TCC_BUBBLE") or 0.0, g("TCC_EA0_RDREQ_32B") or 0.0
            d["FETCH_SIZE_bytes_raw"] = bub * 128 + (g("TCC_EA0_RDREQ") - bub - rd32) * 64 + rd32 * 32
        if g("TCC_EA0_WRREQ") is not None:
            w64 = g("TCC_EA0_WRREQ_64B") or 0.0
            d["WRITE_SIZE_bytes"] = (g("TCC_EA0_WRREQ") - w64) * 32 + w64 * 64
        if "FETCH_SIZE_bytes_raw" in d and "WRITE_SIZE_bytes" in d:
            d["hbm_bytes_per_launch"] = 2 * d["FETCH_SIZE_bytes_raw"] + d["WRITE_SIZE_bytes"]
            d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / t["bytes"] if t["bytes"] else None
            d["hbm_gbs"] = d["hbm_bytes_per_launch"] / (t["us"] * 1e-6) / 1e9
            d["hbm_frac_of_8TBs"] = d["hbm_gbs"] * 1e9 / HBM_PEAK
        if t["bound"] == "hbm":
            d["algorithmic_gbs"] = t["bytes"] / (t["us"] * 1e-6) / 1e9
            d["algorithmic_frac_of_8TBs"] = d["algorithmic_gbs"] * 1e9 / HBM_PEAK
        if g("TCC_HIT") is not None and g("TCC_MISS") is not None and (g("TCC_HIT") + g("TCC_MISS")) > 0:
            d["l2_hit_rate"] = g("TCC_HIT") / (g("TCC_HIT") + g("TCC_MISS"))
        out[tag] = d
    lines = ["| kernel (one steady-state layer's launch) | us | MFMA util (counters) | cyc/MFMA | SQ sampling f | wait / stall / active share | "
             "HBM bytes per launch (x algorithmic) | HBM GB/s (of 8 TB/s) |", "|---|---|---|---|---|---|---|---|"]
    for tag, d in out.items():
        mu = f"{100 * d['mfma_util']:.1f} %" if "mfma_util" in d else "-"
        cp = f"{d['mfma_busy_cycles_per_mfma']:.1f}" if "mfma_busy_cycles_per_mfma" in d else "-"
        sf = f"{d['sq_sampling_factor']:.3f}" if "sq_sampling_factor" in d else "-"
        sh = " / ".join(f"{100 * d[k]:.0f}" for k in ("SQ_WAIT_ANY_share", "SQ_WAIT_INST_ANY_share", "SQ_ACTIVE_INST_ANY_share") if k in d) or "-"
        hb = (f"{d['hbm_bytes_per_launch'] / 1e6:.1f} MB ({d['traffic_over_algorithmic']:.2f}x)" if d.get("traffic_over_algorithmic")
              else "-")
        gb = f"{d['hbm_gbs']:.0f} ({100 * d['hbm_frac_of_8TBs']:.1f} %)" if "hbm_gbs" in d else "-"
        lines.append(f"| `{tag}`: {d['kernel'][:70]} | {d['us_unprofiled']:.1f} | {mu} | {cp} | {sf} | {sh} | {hb} | {gb} |")
    print("\n".join(lines))
    if "--json" in opts:
        json.dump(out, open(opts["--json"], "w"), indent=1)
    if "--md" in opts:
        open(opts["--md"], "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
