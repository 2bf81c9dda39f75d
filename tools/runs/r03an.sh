# last check of the round: smoke + the whole GPU suite + the driver's bench invocation on the final commit
set -u
O=gpurun_out/r03aw; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gputests.log
[ $rc -eq 0 ] || { grep -E "^E |^FAILED" $O/gputests.log | head -20; exit $rc; }
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 - $O/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = r.get("telemetry") or {}
print(r["value"], r["unit"], "ms/step", round(r["ms_per_step"], 2), "roofline", round(r["roofline"]["frac"], 4), "sclk", round(t.get("sclk_mhz_avg", 0)), "power", round(t.get("power_w_avg", 0)))
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in (r.get("extras") or {}).items()})
PY
