mkdir -p gpurun_out/r02f
python -m pytest tests/test_ops_gpu.py tests/test_shipped_sizes_gpu.py -m gpu -x -q -k "gemm or flash or int8 or epilogue or config1 or block" > gpurun_out/r02f/tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02f/tests.log
run() {
  cfg=$1; tag=$2; lib=$3
  LONGLIVE_HIP_LIB=$lib LL_TUNING=$cfg python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r02f/bench_$tag.json 2>gpurun_out/r02f/err.log || { echo "bench failed $tag"; tail -5 gpurun_out/r02f/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02f/bench_$tag.json"))
k={r["tag"]:round(r["avg_us"],1) for r in d["kernels"]["rows"]}
print("$tag", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us", {t:k[t] for t in ("gemm_o","gemm_co","gemm_f2","gemm_cq","flash_attn_cross")})
PY
}
NEW=$PWD/longlive_amd/liblonglive_hip.so; OLD=$PWD/experiments/ab/lib_prev.so
for rep in 1 2; do
run gemm_lds_epi=1 new_$rep $NEW
run gemm_lds_epi=5 new_nopre_$rep $NEW
run gemm_lds_epi=1 old_$rep $OLD
done
