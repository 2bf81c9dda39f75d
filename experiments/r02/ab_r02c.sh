mkdir -p gpurun_out/r02c
run() {
  cfg=$1; tag=$2
  LL_TUNING=$cfg python bench.py --steps 8 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/r02c/bench_$tag.json 2>gpurun_out/r02c/err.log || { echo "bench failed $cfg"; tail -5 gpurun_out/r02c/err.log; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02c/bench_$tag.json"))
print("$cfg", round(d["value"],2), "f/s", round(d["ms_per_step"],2), "ms/step  attn", round(d["roofline"]["avg_us"],1), "us")
PY
}
for rep in 1 2; do
run "gemm_group_m=4" g4_$rep
run "gemm_group_m=2" g2_$rep
run "gemm_group_m=3" g3_$rep
run "gemm_group_m=6" g6_$rep
run "gemm_group_m=8" g8_$rep
run "gemm_group_m=4,gemm_variant_wide=3" g4w3_$rep
run "gemm_group_m=4,gemm_variant_wide=4" g4w4_$rep
done
