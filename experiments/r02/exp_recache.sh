#!/bin/bash
# prompt-switch (recache, M = 18720) latency under the 256x256 GEMM variants
cd /root/repo
for rep in 1 2; do for v in "" "gemm_variant=3" "gemm_variant=4"; do
LL_TUNING=$v timeout -k 10 200 python bench.py --steps 3 --warmup 4 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
print("tuning='$v'", round(d["value"],2), "f/s; switch", round(d["extras"]["prompt_switch_latency"]["value"],2), "ms")
PY
done; done
