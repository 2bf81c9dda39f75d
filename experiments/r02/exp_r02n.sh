mkdir -p gpurun_out/r02n
for rep in 1 2; do for wsv in 0 2; do
LL_TUNING=gemm_ws=$wsv timeout -k 10 200 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02n/b_${wsv}_$rep.json 2>/tmp/b.err || { tail -3 /tmp/b.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r02n/b_${wsv}_$rep.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("gemm_ws=$wsv", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "attn(timed region)", round(d["roofline"]["avg_us"],1), k)
PY
done; done
