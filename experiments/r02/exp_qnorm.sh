#!/bin/bash
# timing experiment (results invalid): the cross-attention q RMSNorm launch skipped = upper bound of fusing it into its neighbours
cd /root/repo
for rep in 1 2 3; do for v in 0 1; do
LL_EXP_SKIP_QNORM=$v timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
print("skip_qnorm=$v", round(d["value"],2), "f/s", round(d["ms_per_step"],2))
PY
done; done
