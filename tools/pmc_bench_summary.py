#!/usr/bin/env python3
"""Summary of ONE `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -- python3 bench.py ...` pass
(tools/gpu_batch.sh pmc_bench): per kernel, over the LAST third of its dispatches (the steady-state blocks), mean duration of the
counted dispatch, MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) and the implied clock.
Same definitions as tools/pmc_inpipe_summary.py, so the figures sit next to the kbench-layerseq ones.

    python tools/pmc_bench_summary.py <counter_collection.csv>
"""
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_inpipe_summary import label, short  # noqa: E402


def main(path):
    per = defaultdict(dict)
    name = {}
    for r in csv.DictReader(open(path)):
        i = int(r["Dispatch_Id"])
        name[i] = short(r["Kernel_Name"])
        per[i][r["Counter_Name"]] = per[i].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if r.get("End_Timestamp") and r.get("Start_Timestamp"):
            per[i]["_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    ids = sorted(per)
    steady = ids[len(ids) * 2 // 3:]
    groups = defaultdict(list)
    for i in steady:
        groups[label(name[i], per[i].get("_us"))].append(per[i])
    rows = []
    for k, ds in groups.items():
        us = sum(d.get("_us", 0.0) for d in ds) / len(ds)
        cyc = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in ds) / len(ds) / 8.0
        busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in ds) / len(ds)
        insts = sum(d.get("SQ_INSTS_MFMA", 0.0) for d in ds) / len(ds)
        rows.append((us * len(ds), k, len(ds), us, cyc, busy / (1024.0 * cyc) if cyc else 0.0, cyc / us / 1e3 if us else 0.0, insts))
    rows.sort(reverse=True)
    print(f"rocprofv3 --pmc over bench.py itself: {len(ids)} dispatches, steady-state third = {len(steady)}\n")
    print("| kernel | dispatches | avg us (counted run) | MFMA util | implied GHz | MFMA insts / dispatch |")
    print("|---|---|---|---|---|---|")
    for _, k, n, us, cyc, util, ghz, insts in rows[:16]:
        print(f"| `{k[:70]}` | {n} | {us:.1f} | {100 * util:.1f} % | {ghz:.2f} | {insts:.3g} |")


if __name__ == "__main__":
    main(sys.argv[1])
