set -u
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; rc=$?; tail -4 $O/gputests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/gputests.log | head -20; exit $rc; }
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json; r=json.load(open('gpurun_out/r03d/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_us'], r['roofline']['frac'], r['roofline']['kernel'][:60]); print(r['telemetry']); print({k:(v.get('value', v) if isinstance(v,dict) else v) for k,v in r['extras'].items()})
for k in r['kernels']['rows'][:12]: print("%-22s %8.1f us x%4d  frac %.3f" % (k['tag'], k['avg_us'], k['launches'], k['frac']))
PY
./tools/kbench attnx 20 2 > $O/kbench_attnx.txt 2>&1; cat $O/kbench_attnx.txt
