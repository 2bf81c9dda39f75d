set -u
O=gpurun_out/r03h; mkdir -p $O
timeout -k 5 60 ./tools/attn_asm_diag_x0 4680 18720 20 2>&1 | grep -E "diag build|wave 0" | sed 's/(median.*//'
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "asm_kernel or flash_attn" > $O/tests_asm.log 2>&1; rc=$?; tail -3 $O/tests_asm.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests_asm.log | head; exit $rc; }
timeout -k 10 300 python -m pytest "tests/test_shipped_sizes_gpu.py::test_flash_attn_production_grids" -m gpu -x -q -s > $O/tests_grid.log 2>&1; rc=$?; tail -3 $O/tests_grid.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests_grid.log | head; exit $rc; }
./tools/kbench attnx 20 2 2>&1 | grep -v "asm-g" | tee $O/kbench_attnx.txt
B="timeout -k 10 300 python bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline"
for r in 1 2; do
  LONGLIVE_HIP_LIB=$PWD/longlive_amd/liblonglive_hip_prev.so $B > $O/ab_prev_$r.json 2>> $O/ab.err || exit 1
  $B > $O/ab_new_$r.json 2>> $O/ab.err || exit 1
done
python - <<'PY'
import json
for n in ("ab_prev_1","ab_new_1","ab_prev_2","ab_new_2"):
    r=json.load(open(f"gpurun_out/r03h/{n}.json")); t=r.get("telemetry") or {}
    print(n, "%.2f f/s" % r["value"], "attn %.1f us" % r["roofline"]["avg_us"], "sclk %.0f W %.0f" % (t.get("sclk_mhz_avg") or 0, t.get("power_w_avg") or 0))
PY
