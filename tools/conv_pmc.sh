#!/bin/bash
# rocprofv3 counter passes over tools/conv_pmc_run.py (no tracing flags beside --pmc).  Run ON the GPU box: bash tools/conv_pmc.sh <outdir>
set -u
OUT=${1:-gpurun_out/convpmc}
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
python3 tools/conv_pmc_run.py 20 > "$OUT/work.log" 2>&1 || { tail -5 "$OUT/work.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/conv_pmc_run.py 4 > "$OUT/$name.log" 2>&1
  rc=$?
  if [ $rc -ge 124 ]; then echo "pass $name killed (rc $rc): stopping"; exit $rc; fi
  if [ $rc -ne 0 ]; then echo "pass $name failed (rc $rc), skipped"; tail -3 "$OUT/$name.log"; else echo "pass $name ok"; fi
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass tcc_rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum GRBM_GUI_ACTIVE
pass tcc_wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
work = json.loads([l for l in open(os.path.join(out, "work.log")) if l.startswith("#WORK ")][0][6:])
c = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    rows = {}
    for r in csv.DictReader(open(f)):
        if "conv_halo" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        rows.setdefault(d, {})
        rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ds = sorted(rows)[2:]                       # drop the two warm-up launches
    for k in set().union(*[set(rows[d]) for d in ds]):
        c[k] = sum(rows[d].get(k, 0.0) for d in ds) / len(ds)
g = lambda n: c.get(n, c.get(n + "_sum"))
res = dict(work)
cyc = g("GRBM_GUI_ACTIVE") / 8
need = work["flop"] / 16384.0
f = g("SQ_INSTS_MFMA") / need
res.update(kernel_cycles_profiled=cyc, sq_sampling_factor=f, mfma_busy_cycles_per_mfma=g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_INSTS_MFMA"),
           mfma_util=g("SQ_VALU_MFMA_BUSY_CYCLES") / f / (1024 * cyc), achieved_tflops=work["flop"] / work["us"] / 1e6)
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
    res[k + "_share"] = g(k) / g("SQ_WAVE_CYCLES")
if g("SQ_LDS_BANK_CONFLICT") is not None:
    res["lds_bank_conflict_share"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
bub, rd32 = g("TCC_BUBBLE") or 0.0, g("TCC_EA0_RDREQ_32B") or 0.0
fetch = bub * 128 + (g("TCC_EA0_RDREQ") - bub - rd32) * 64 + rd32 * 32
w64 = g("TCC_EA0_WRREQ_64B") or 0.0
write = (g("TCC_EA0_WRREQ") - w64) * 32 + w64 * 64
res.update(hbm_bytes_per_launch=2 * fetch + write, traffic_over_algorithmic=(2 * fetch + write) / work["bytes"],
           hbm_gbs=(2 * fetch + write) / work["us"] / 1e3)
json.dump(res, open(os.path.join(out, "conv_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
