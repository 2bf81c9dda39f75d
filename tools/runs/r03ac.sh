# full GPU suite + driver-style bench with the generated GEMM kernels as the default (gemm_asm = 3)
set -u
O=gpurun_out/r03ac; mkdir -p $O
echo "== gpu tests"; timeout -k 10 1000 python -m pytest ${GPU_TESTS:-tests} -m gpu -q -x > $O/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/gputests.log
[ $rc -eq 0 ] || { grep -E "^E |Error|FAILED" $O/gputests.log | head -30; exit $rc; }
echo "== bench --steps 20 --warmup 5"; timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - $O/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["value"], r["unit"], "ms/step", r["ms_per_step"], "roofline", r["roofline"])
t = r.get("telemetry") or {}
print({k: v for k, v in t.items() if k != "gpu_metrics_delta"})
for k in (r.get("extras") or {}).get("kernels", [])[:8]:
    print("  ", k.get("tag"), k.get("kernel", "")[:60], round(k.get("avg_us", 0), 1), round(k.get("frac", 0), 3))
PY
