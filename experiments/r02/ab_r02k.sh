mkdir -p gpurun_out/r02k
python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_shipped_sizes_gpu.py tests/test_boundary_gpu.py -m gpu -x -q > gpurun_out/r02k/tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02k/tests.log
run() {
  tag=$1; fv=$2
  LL_FUSE_V=$fv python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02k/bench_$tag.json 2>gpurun_out/r02k/err.log || { echo "bench failed $tag"; tail -5 gpurun_out/r02k/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02k/bench_$tag.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$tag", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us", {t:k[t] for t in ("gemm_qkv","qk_norm_rope_kv_store","kv_roll")}, "int8", round(d["extras"]["int8_w8a8"]["value"],2))
PY
}
for rep in 1 2; do
run fuse_$rep 1
run nofuse_$rep 0
done
