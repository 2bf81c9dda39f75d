for rep in 1 2; do for cfg in "attn_pp_min_keys=1024" "attn_pp_min_keys=512"; do
LL_TUNING=$cfg timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$cfg", round(d["value"],2), "f/s", round(d["ms_per_step"],2), {t:k[t] for t in ("flash_attn_self","flash_attn_cross")})
PY
done; done
