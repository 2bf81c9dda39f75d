#!/bin/bash
# full GPU suite + smoke + bench + e2e on the current build
cd /root/repo
O=gpurun_out/full; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/full/bench.json'))
print({k:d[k] for k in ('metric','value','ms_per_step','vs_baseline')}, d['roofline'], {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d['extras'].items()})
PY
timeout -k 10 500 python tools/run_configs.py 60 --vae --e2e-only > $O/e2e.json 2> $O/e2e.err || { tail -5 $O/e2e.err; exit 1; }
cat $O/e2e.json
