#!/bin/bash
# spread LDS-DMA pieces: parity tests of attention, then tile-cycle diagnostics
cd /root/repo
O=gpurun_out/r02u; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "attn or attention" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
for b in 1 8 4 2; do
  timeout -k 5 60 ./tools/attn_diag_$b 2 >> $O/diag.txt 2>&1 || { echo "diag $b failed" >> $O/diag.txt; exit 1; }
done
done
cat $O/diag.txt
