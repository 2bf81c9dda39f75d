#!/bin/bash
# pipeline timing experiment (results invalid): FFN2 emulated as 256x256 x split-K 2 (v3 / v4), no reduction
cd /root/repo
for rep in 1 2; do for v in 0 3 4; do
LL_EXP_FFN2=$v timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("exp=$v", round(d["value"],2), "f/s", round(d["ms_per_step"],2), {t:k.get(t) for t in ("flash_attn_self","gemm_f2","gemm_f1","gemm_qkv","gemm_o")})
PY
done; done
