#!/bin/bash
cd /root/repo
O=gpurun_out/splitk5; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "splitk" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
./tools/kbench shipped 30 | grep -E "^gemm_f2"
LL_TUNING=gemm_splitk_l2=0 ./tools/kbench shipped 30 | grep -E "^gemm_f2" || true
for rep in 1 2 3; do for v in "LL_SPLITK=0" "LL_SPLITK=1 LL_TUNING=gemm_splitk_l2=0" "LL_SPLITK=1"; do
env $v timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$v", round(d["value"],2), "f/s", round(d["ms_per_step"],2), {t:k.get(t) for t in ("flash_attn_self","gemm_f2","gemm_f1")})
PY
done; done
