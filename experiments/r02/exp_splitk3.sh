#!/bin/bash
cd /root/repo
O=gpurun_out/splitk3; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "splitk" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2 3; do for v in 0 1; do
LL_SPLITK=$v timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --quant int8 > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("int8 splitk=$v", round(d["value"],2), "f/s", round(d["ms_per_step"],2), {t:k.get(t) for t in ("flash_attn_self","gemm_f2","gemm_f1","gemm_qkv","gemm_o")})
PY
done; done
