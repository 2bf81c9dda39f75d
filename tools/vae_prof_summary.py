"""Per-kernel table of the timed region of tools/vae_bench.py from a rocprofv3 --kernel-trace rocpd database.
usage: python tools/vae_prof_summary.py gpurun_out/vaeprof/vae_results.db > profiles/rNN_vae_kernels.md"""
import re
import sqlite3
import sys
from collections import defaultdict


def main():
    c = sqlite3.connect(sys.argv[1]).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    last = [i for i, r in enumerate(rows) if "vae_unscale" in r[0]][-1]      # the timed decode_to_pixel call
    sel = rows[last:]
    agg = defaultdict(lambda: [0, 0.0])
    for n, s, e in sel:
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        m = re.match(r"_Z14conv_cl_kernelILi(\d)ELi(\d)ELi(\d)EE", n)
        if m:
            n = f"conv_cl_kernel<EPI={m.group(1)}, NT={m.group(2)}, MODE={m.group(3)}>"
        agg[n[:72]][0] += 1
        agg[n[:72]][1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    span = (sel[-1][2] - sel[0][1]) / 1e3
    print(f"timed region: {span / 1e3:.2f} ms wall, {tot / 1e3:.2f} ms of kernels ({100 * tot / span:.1f} % busy), {len(sel)} launches\n")
    print("| kernel | launches | total ms | avg us | share |\n|---|---|---|---|---|")
    for n, (k, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:20]:
        print(f"| `{n}` | {k} | {t / 1e3:.3f} | {t / k:.1f} | {100 * t / tot:.1f} % |")


if __name__ == "__main__":
    main()
