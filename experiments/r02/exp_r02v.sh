#!/bin/bash
# LDS-DMA placement inside the softmax phase: head (0) vs tail (1)
cd /root/repo
O=gpurun_out/r02v; mkdir -p $O
for rep in 1 2; do
for v in 1_at0 1_at1 8_at0 8_at1; do
  echo -n "$v " >> $O/diag.txt
  timeout -k 5 60 ./tools/attn_diag_$v 2 >> $O/diag.txt 2>&1 || { echo "diag $v failed" >> $O/diag.txt; exit 1; }
done
done
cat $O/diag.txt
