mkdir -p gpurun_out/r02p
for rep in 1 2; do for cfg in "gemm_ws=0" "gemm_ws=2,gemm_ws_mask=1" "gemm_ws=2,gemm_ws_mask=3" "gemm_ws=2,gemm_ws_mask=12" "gemm_ws=2,gemm_ws_mask=4"; do
LL_TUNING=$cfg timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02p/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02p/b.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$cfg", round(d["value"],2), "f/s", round(d["ms_per_step"],2), {t:k[t] for t in ("flash_attn_self","gemm_f2","gemm_f1","gemm_qkv","gemm_o")})
PY
done; done
