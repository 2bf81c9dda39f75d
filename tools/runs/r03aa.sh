set -u
O=gpurun_out/r03aa; mkdir -p $O
export TMPDIR=/tmp
for t in gemm_asm=0 gemm_variant=2 gemm_variant=3 gemm_asm=3 gemm_asm=0; do
  LL_TUNING=$t timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$t -- ./tools/kbench layerseq 600 > $O/t_$t.log 2>&1 || { echo "trace $t failed"; tail -3 $O/t_$t.log; exit 1; }
  f=$(find $O/t_$t -name "*kernel_stats.csv" | head -1)
  echo "== $t ($(grep layerseq: $O/t_$t.log))"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
done 2>&1 | tee $O/summary.txt
find $O -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $O -name "*.db" -delete
