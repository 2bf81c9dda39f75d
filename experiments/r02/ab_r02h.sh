mkdir -p gpurun_out/r02h
LL_TUNING_TEST=gemm_stagger=1 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm or int8 or epilogue or modulation" > gpurun_out/r02h/tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02h/tests.log
run() {
  cfg=$1; tag=$2
  LL_TUNING=$cfg python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02h/bench_$tag.json 2>gpurun_out/r02h/err.log || { echo "bench failed $tag"; tail -5 gpurun_out/r02h/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02h/bench_$tag.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$tag", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us", {t:k[t] for t in ("gemm_o","gemm_co","gemm_cq","gemm_f2","gemm_qkv","gemm_f1")})
PY
}
for rep in 1 2; do
run gemm_stagger=1 stag_$rep
run gemm_stagger=0 base_$rep
done
