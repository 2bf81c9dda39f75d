set -u
O=gpurun_out/r03b; mkdir -p $O
python -m pytest tests/test_ops_gpu.py::test_gemm_splitk_handoff_is_fail_safe tests/test_ops_gpu.py::test_gemm_splitk_l2_exchange_matches tests/test_ops_gpu.py::test_gemm_w8a8_splitk_is_exact "tests/test_shipped_sizes_gpu.py::test_flash_attn_production_grids" -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 10 --warmup 4 --no-extras --no-cpu-baseline"
for r in 1 2; do
  $B > $O/ab_base_$r.json 2>> $O/ab.err || exit 1
  LL_TUNING=attn_mfma16=1 $B > $O/ab_m16_$r.json 2>> $O/ab.err || exit 1
done
python - <<'PY'
import json
for n in ("ab_base_1","ab_m16_1","ab_base_2","ab_m16_2"):
    r=json.load(open(f"gpurun_out/r03b/{n}.json")); t=r.get("telemetry") or {}
    print(n, "%.2f f/s" % r["value"], "attn %.1f us" % r["roofline"]["avg_us"], r["roofline"]["kernel"][:40], "sclk", t.get("sclk_mhz_avg"), "W", t.get("power_w_avg"))
PY
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json; r=json.load(open('gpurun_out/r03b/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_us'], r['roofline']['frac']); print(r['telemetry']); print({k:(v.get('value', v) if isinstance(v,dict) else v) for k,v in r['extras'].items()})
PY
bash tools/pmc_inpipe.sh $O/pmc && python tools/pmc_inpipe_summary.py $O/pmc --md $O/pmc_inpipe.md --json $O/pmc_inpipe.json > /dev/null; head -25 $O/pmc_inpipe.md | cut -c1-220
find $O/pmc -name "*.csv" -size +1M -delete
