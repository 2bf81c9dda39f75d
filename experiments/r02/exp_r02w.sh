#!/bin/bash
# softmax phase: packed fp32 VALU (pk) and the cost of the exponentials (diag 32 = exp2 replaced by identity)
cd /root/repo
O=gpurun_out/r02w; mkdir -p $O
for rep in 1 2; do
for v in 1_pk0 1_pk1 10_pk0 10_pk1 33_pk0 33_pk1 42_pk0 42_pk1; do
  echo -n "$v " >> $O/diag.txt
  timeout -k 5 60 ./tools/attn_diag_$v 2 >> $O/diag.txt 2>&1 || { echo "diag $v failed" >> $O/diag.txt; exit 1; }
done
done
cut -c1-200 $O/diag.txt
