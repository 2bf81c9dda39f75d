#!/bin/bash
# runs every tools/attn_diag_* variant twice, interleaved (in-kernel clock + cycles per key tile); output gpurun_out/diag/diag.txt
cd /root/repo
O=gpurun_out/diag; mkdir -p $O; : > $O/diag.txt
for rep in 1 2; do
for f in tools/attn_diag_*; do
  v=${f#tools/attn_diag_}
  echo -n "$v " >> $O/diag.txt
  timeout -k 5 60 $f ${DIAG_SECS:-2} >> $O/diag.txt 2>&1 || { echo "diag $v failed" >> $O/diag.txt; cat $O/diag.txt; exit 1; }
done
done
python3 - <<'PY'
import json,collections
r=collections.defaultdict(list)
for l in open('gpurun_out/diag/diag.txt'):
    v,_,j=l.partition(' ')
    try: d=json.loads(j)
    except Exception: print(l.strip()); continue
    r[v].append(d)
for v,ds in r.items():
    print(f"{v:12s} us " + "/".join(f"{d['us_per_launch']:.1f}" for d in ds) + "  cyc/tile " + "/".join(f"{d['cycles_per_tile_median']:.0f}" for d in ds) + "  GHz " + "/".join(f"{d['clock_ghz_median']:.3f}" for d in ds))
PY
