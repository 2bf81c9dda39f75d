set -u
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "asm_kernel" > $O/tests_asm.log 2>&1; rc=$?; tail -5 $O/tests_asm.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests_asm.log | head -20; exit $rc; }
timeout -k 10 400 python -m pytest "tests/test_shipped_sizes_gpu.py::test_flash_attn_production_grids" -m gpu -x -q -s > $O/tests_grid.log 2>&1; rc=$?; tail -4 $O/tests_grid.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests_grid.log | head -20; exit $rc; }
B="timeout -k 10 300 python bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline"
for r in 1 2; do
  $B > $O/ab_base_$r.json 2>> $O/ab.err || exit 1
  LL_TUNING=attn_asm=1 $B > $O/ab_asm1_$r.json 2>> $O/ab.err || exit 1
  LL_TUNING=attn_asm=2 $B > $O/ab_asm2_$r.json 2>> $O/ab.err || exit 1
done
python - <<'PY'
import json
for n in ("ab_base_1","ab_asm1_1","ab_asm2_1","ab_base_2","ab_asm1_2","ab_asm2_2"):
    r=json.load(open(f"gpurun_out/r03c/{n}.json")); t=r.get("telemetry") or {}
    print(n, "%.2f f/s" % r["value"], "attn %.1f us" % r["roofline"]["avg_us"], r["roofline"]["kernel"][:28], "sclk %.0f W %.0f" % (t.get("sclk_mhz_avg") or 0, t.get("power_w_avg") or 0))
PY
