mkdir -p gpurun_out/r02g
python -m pytest tests -m gpu -x -q > gpurun_out/r02g/tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02g/tests.log
run() {
  tag=$1; mt=$2
  LL_MODTAB=$mt python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02g/bench_$tag.json 2>gpurun_out/r02g/err.log || { echo "bench failed $tag"; tail -5 gpurun_out/r02g/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02g/bench_$tag.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$tag", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us", {t:k[t] for t in ("ln_modulate","gemm_o","gemm_f2","layernorm_affine","rmsnorm")})
PY
}
for rep in 1 2; do
run tab_$rep 1
run notab_$rep 0
done
