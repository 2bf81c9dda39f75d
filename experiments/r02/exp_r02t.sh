#!/bin/bash
# attention tile-cycle diagnostics (tools/attn_diag.hip): in-kernel clock + cycles per key tile with parts of the loop removed
cd /root/repo
O=gpurun_out/r02t; mkdir -p $O
for rep in 1 2; do
for b in 1 2 16 4 8 6 10 12; do
  timeout -k 5 60 ./tools/attn_diag_$b 2 >> $O/diag.txt 2>&1 || { echo "diag $b failed" >> $O/diag.txt; exit 1; }
done
done
cat $O/diag.txt
