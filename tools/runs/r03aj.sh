set -u
O=gpurun_out/r03aj; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t5 -- python3 tools/t5_bench.py 5 > $O/t5.log 2>&1; echo "rc=$?"; tail -2 $O/t5.log | cut -c1-300
f=$(find $O/t5 -name "*kernel_stats.csv" | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
