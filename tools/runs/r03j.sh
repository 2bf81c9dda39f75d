set -u
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; rc=$?; tail -4 $O/gputests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/gputests.log | head -20; exit $rc; }
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<'PY'
import json; r=json.load(open('gpurun_out/r03j/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['avg_us'], r['roofline']['frac']); print({k:(v.get('value', v) if isinstance(v,dict) else v) for k,v in r['extras'].items()})
PY
bash tools/pmc_inpipe.sh $O/pmc; python tools/pmc_inpipe_summary.py $O/pmc --md $O/pmc_inpipe.md --json $O/pmc_inpipe.json > /dev/null 2> $O/pmc_summary.err; head -25 $O/pmc_inpipe.md | cut -c1-230
find $O/pmc -name "*.csv" -size +1M -delete
