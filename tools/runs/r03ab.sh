set -u
O=gpurun_out/r03ab; mkdir -p $O
for i in 1 2 3; do
  for t in 0 3; do
    LL_TUNING=gemm_asm=$t timeout -k 10 400 python3 bench.py --steps 14 --warmup 4 --no-extras --no-cpu-baseline --no-kernel-timer > $O/bench_${t}_$i.json 2> $O/bench_${t}_$i.err || { echo "bench failed"; tail -5 $O/bench_${t}_$i.err; exit 1; }
    python3 - $O/bench_${t}_$i.json $t <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
tel = r.get("telemetry") or {}
print(f"gemm_asm={sys.argv[2]}: {r['value']:.2f} {r['unit']}  ms/step {r['ms_per_step']:.1f}  telemetry {json.dumps(tel)[:300]}")
PY
  done
done 2>&1 | tee $O/summary.txt
