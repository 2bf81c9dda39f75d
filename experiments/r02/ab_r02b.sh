mkdir -p gpurun_out/r02b
python -m pytest tests/test_ops_gpu.py tests/test_shipped_sizes_gpu.py -m gpu -x -q -s -k "flash or qk_norm or config1 or gemm" > gpurun_out/r02b/tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02b/tests.log
for rep in 1 2; do
for cfg in "attn_sk_wgs=0,gemm_group_m=4" "attn_sk_wgs=-1,gemm_group_m=4" "attn_sk_wgs=0,gemm_group_m=1" "attn_sk_wgs=-1,gemm_group_m=1"; do
  LL_TUNING=$cfg python bench.py --steps 8 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r02b/bench_${cfg//[=,]/_}_$rep.json 2>gpurun_out/r02b/err.log || { echo "bench failed $cfg"; tail -5 gpurun_out/r02b/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02b/bench_${cfg//[=,]/_}_$rep.json"))
print("$cfg", "rep$rep", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us")
PY
done; done
