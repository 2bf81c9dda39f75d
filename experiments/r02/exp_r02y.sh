#!/bin/bash
# pipeline A/B: softmax phase at priority 1 (lib_p2.so, -DLL_ATTN_PRIO=2) vs shipped (matrix phase at priority 1)
cd /root/repo
for rep in 1 2 3; do for lib in longlive_amd/liblonglive_hip.so experiments/ab/lib_p2.so; do
LONGLIVE_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$lib", round(d["value"],2), "f/s", round(d["ms_per_step"],2), {t:k[t] for t in ("flash_attn_self","flash_attn_cross")})
PY
done; done
