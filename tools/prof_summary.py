#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace CSV of `bench.py` to a per-kernel table of the LAST `--blocks` AR blocks
(steady state: full 18720-slot window) plus the GPU-busy fraction of that window.

    python tools/prof_summary.py gpurun_out/prof/.../NNN_kernel_trace.csv --blocks 2 > profiles/r01_kernel_summary.md

A block = 5 DiT forwards x 30 layers; self- and cross-attention are each launched once per layer per forward, so the
last blocks*300 flash_attn launches delimit the window."""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name):
    n = re.sub(r"\(.*", "", name).replace("void ", "")
    n = re.sub(r"at::native::", "", n)
    return n[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--layers", type=int, default=30)
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["d"] = r["e"] - r["s"]
    rows.sort(key=lambda r: r["s"])
    att = [r for r in rows if "flash_attn" in r["Kernel_Name"]]
    if not att:
        sys.exit("no flash_attn launches in trace")
    per_block = 5 * args.layers * 2                    # self + cross launches per block
    window = att[-args.blocks * per_block:]
    t0 = window[0]["s"]
    # the forward's kernels that precede its first attention launch (patch/time embed, LN-modulate, QKV GEMM, roll ...)
    prev = [r for r in rows if r["s"] < t0][-12:]
    lead = [r["s"] for r in prev if "patchify" in r["Kernel_Name"]]
    if lead:
        t0 = lead[-1]
    t1 = max(r["e"] for r in rows)
    sel = [r for r in rows if r["s"] >= t0]
    wall = t1 - t0
    busy = sum(r["d"] for r in sel)
    groups = defaultdict(list)
    for r in sel:
        key = short(r["Kernel_Name"])
        if "flash_attn" in key:
            key += " (self, Lk=18720)" if r["d"] > 200000 else " (cross, Lk=512)"
        if "gemm_bf16_kernel" in key:
            gx = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
            key = f"{key} wgs={gx}"
        groups[key].append(r["d"])
    print(f"steady-state window: last {args.blocks} AR blocks = {wall/1e6:.2f} ms wall, "
          f"{busy/1e6:.2f} ms of kernels ({100*busy/wall:.1f}% GPU-busy), {len(sel)} launches, "
          f"{wall/1e6/args.blocks:.2f} ms/block\n")
    print("| kernel | launches | avg us | total ms | % of window |")
    print("|---|---:|---:|---:|---:|")
    for key, ds in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        tot = sum(ds)
        if tot / wall < 0.0005:
            continue
        print(f"| {key} | {len(ds)} | {tot/len(ds)/1e3:.1f} | {tot/1e6:.2f} | {100*tot/wall:.2f} |")
    if busy <= wall:
        print(f"| (idle / launch gaps) | | | {(wall-busy)/1e6:.2f} | {100*(wall-busy)/wall:.2f} |")
    else:       # two HIP streams (the clean-context forward beside the next block's first forward): kernel time exceeds wall time
        print(f"| (kernels of the two streams running side by side: kernel time beyond the wall time) | | | {(busy-wall)/1e6:.2f} | {100*(busy-wall)/wall:.2f} |")
        print("\nDurations of launches that ran beside the other stream's kernels include the time they waited for CUs (one workgroup of a generated "
              "kernel owns a CU's registers): e.g. row kernels and the 512-key attention stretch when a self-attention launch of the other stream holds 228 CUs.")


if __name__ == "__main__":
    main()
