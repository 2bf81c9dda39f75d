set -u
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests/test_ops_gpu.py tests/test_abi.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json; r=json.load(open('gpurun_out/r03a/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_us'], r['roofline']['frac']); print(r['telemetry']); print({k:(v.get('value') if isinstance(v,dict) else v) for k,v in r['extras'].items()})
PY
bash tools/pmc_inpipe.sh $O/pmc && python tools/pmc_inpipe_summary.py $O/pmc --md $O/pmc_inpipe.md --json $O/pmc_inpipe.json > /dev/null; head -25 $O/pmc_inpipe.md | cut -c1-220
find $O/pmc -name "*.csv" -size +1M -delete
