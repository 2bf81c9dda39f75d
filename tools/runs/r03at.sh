set -u
O=gpurun_out/r03at; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "w8a8" > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E |^FAILED" $O/tests.log | head -20; exit $rc; }
for t in 3 19 3 19; do
  LL_TUNING=gemm_asm=$t timeout -k 10 400 python3 bench.py --steps 10 --warmup 4 --quant int8 --no-extras --no-cpu-baseline --no-kernel-timer > $O/b_$t.json 2> $O/b_$t.err; rc=$?; [ $rc -eq 0 ] || { tail -3 $O/b_$t.err; exit $rc; }
  python3 - $O/b_$t.json $t <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = r.get("telemetry") or {}
print(f"int8, gemm_asm={sys.argv[2]}: {r['value']:.2f} {r['unit']}  ms/step {r['ms_per_step']:.1f}  sclk {t.get('sclk_mhz_avg', 0):.0f}  power {t.get('power_w_avg', 0):.0f}")
PY
done | tee $O/ab_int8.txt
