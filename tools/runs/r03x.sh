set -u
O=gpurun_out/r03x; mkdir -p $O
export TMPDIR=/tmp
for t in 0 1 3 0; do
  LL_TUNING=gemm_asm=$t ./tools/kbench layerseq 300 | tee -a $O/layerseq.txt
done
for t in 0 1 3; do
  LL_TUNING=gemm_asm=$t timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$t -- ./tools/kbench layerseq 300 > $O/t$t.log 2>&1 || { echo "trace t$t failed"; tail -3 $O/t$t.log; exit 1; }
  f=$(find $O/t$t -name "*kernel_stats.csv" | head -1)
  echo "== gemm_asm=$t ($(grep layerseq $O/t$t.log))"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
done 2>&1 | tee $O/summary.txt
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
