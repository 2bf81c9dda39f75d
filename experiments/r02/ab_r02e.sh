mkdir -p gpurun_out/r02e
python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py tests/test_checkpoint.py -m gpu -x -q -k "gemm or int8 or w8a8 or epilogue or patchify or encoder or lora or t5" > gpurun_out/r02e/tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02e/tests.log
run() {
  cfg=$1; tag=$2
  LL_TUNING=$cfg python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02e/bench_$tag.json 2>gpurun_out/r02e/err.log || { echo "bench failed $tag"; tail -5 gpurun_out/r02e/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02e/bench_$tag.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$tag", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us", {t:k[t] for t in ("gemm_o","gemm_co","gemm_f2","gemm_cq","gemm_qkv","gemm_f1")}, "int8", round(d["extras"]["int8_w8a8"]["value"],2))
PY
}
for rep in 1 2; do
run gemm_lds_epi=1 lds_$rep
run gemm_lds_epi=0 reg_$rep
done
