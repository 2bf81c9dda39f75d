#!/bin/bash
# segment-schedule attention (attn_variant=3): parity tests under the variant, then time/energy against variant 2
cd /root/repo
O=gpurun_out/r02x; mkdir -p $O
LL_TUNING_TEST=attn_variant=3 timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -x -k "attn or attention" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for rep in 1 2; do
  for v in 2 3; do timeout -k 5 60 ./tools/kenergy attn $v 2 >> $O/kenergy.txt 2>&1 || exit 1; done
done
cat $O/kenergy.txt
