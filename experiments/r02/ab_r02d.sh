mkdir -p gpurun_out/r02d
python -m pytest tests/test_ops_gpu.py tests/test_vae_gpu.py -m gpu -x -q -k "gemm or conv or int8 or w8a8 or epilogue or decode" > gpurun_out/r02d/tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02d/tests.log
run() {
  lib=$1; tag=$2
  LONGLIVE_HIP_LIB=$lib python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02d/bench_$tag.json 2>gpurun_out/r02d/err.log || { echo "bench failed $tag"; tail -5 gpurun_out/r02d/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02d/bench_$tag.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$tag", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us", {t:k[t] for t in ("gemm_o","gemm_co","gemm_f2","gemm_cq","gemm_qkv","gemm_f1")})
PY
}
for rep in 1 2; do
run $PWD/longlive_amd/liblonglive_hip.so new_$rep
run $PWD/experiments/ab/lib_prev.so prev_$rep
done
