#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files: per kernel, per-launch average of every counter, plus the
derived quantities used in DESIGN.md / bench.py's roofline.traffic:

  FETCH_SIZE [KB]  = (TCC_BUBBLE*128 + (TCC_EA0_RDREQ - TCC_BUBBLE - TCC_EA0_RDREQ_32B)*64 + TCC_EA0_RDREQ_32B*32)/1024
  WRITE_SIZE [KB]  = ((TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B)*32 + TCC_EA0_WRREQ_64B*64)/1024
  hbm_bytes        = 2 * FETCH_SIZE*1024 + WRITE_SIZE*1024      (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports
                     exactly 1/2 of the bytes of a wide coalesced streaming read; WRITE_SIZE is exact)
  MfmaUtil [%]     = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100   (ROCm's gfx94x formula)

(the derived FETCH_SIZE / WRITE_SIZE counters themselves crash rocprofv3 7.2 on this box, so their base counters are
collected and the published expressions are evaluated here).

    python tools/pmc_summary.py dir_or_csv [dir_or_csv ...] [--json out.json]
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "")[:48]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    jout = None
    if "--json" in sys.argv:
        jout = sys.argv[sys.argv.index("--json") + 1]
        args = [a for a in args if a != jout]
    files = []
    for a in args:
        files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
    acc = defaultdict(lambda: defaultdict(float))       # kernel -> counter -> sum over launches
    disp = defaultdict(lambda: defaultdict(set))        # kernel -> counter -> dispatch ids
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if "Grid_Size" in r and "flash_attn" in k:
                k += f" grid={r['Grid_Size']}"
            c = r["Counter_Name"]
            acc[k][c] += float(r["Counter_Value"])
            disp[k][c].add((f, r["Dispatch_Id"]))
    out = {}
    for k in sorted(acc):
        per = {c: acc[k][c] / max(1, len(disp[k][c])) for c in acc[k]}
        g = lambda n: per.get(n, per.get(n + "_sum"))
        d = dict(per)
        if g("TCC_EA0_RDREQ") is not None:
            bub, rd32 = g("TCC_BUBBLE") or 0.0, g("TCC_EA0_RDREQ_32B") or 0.0
            d["FETCH_SIZE_KB"] = (bub * 128 + (g("TCC_EA0_RDREQ") - bub - rd32) * 64 + rd32 * 32) / 1024
        if g("TCC_EA0_WRREQ") is not None:
            w64 = g("TCC_EA0_WRREQ_64B") or 0.0
            d["WRITE_SIZE_KB"] = ((g("TCC_EA0_WRREQ") - w64) * 32 + w64 * 64) / 1024
        if "FETCH_SIZE_KB" in d or "WRITE_SIZE_KB" in d:
            d["hbm_bytes_per_launch"] = 2 * d.get("FETCH_SIZE_KB", 0) * 1024 + d.get("WRITE_SIZE_KB", 0) * 1024
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("GRBM_GUI_ACTIVE"):
            d["MfmaUtil_pct"] = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") * 4)
        d["launches"] = max(len(v) for v in disp[k].values())
        out[k] = d
    for k, d in out.items():
        print(k)
        for c in sorted(d):
            print(f"    {c:32s} {d[c]:.6g}")
    if jout:
        json.dump(out, open(jout, "w"), indent=1)


if __name__ == "__main__":
    main()
