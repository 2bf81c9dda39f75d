#!/bin/bash
# halo-tile convolution: parity tests, then the decoder bench with and without it
cd /root/repo
O=gpurun_out/vae; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_vae_gpu.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for cfg in conv_halo=1 conv_halo=0 conv_halo=1 conv_halo=0; do
  echo -n "$cfg " >> $O/bench.txt
  LL_TUNING=$cfg timeout -k 10 200 python tools/vae_bench.py 9 2 >> $O/bench.txt 2>$O/bench.err || { tail -5 $O/bench.err; exit 1; }
done
cat $O/bench.txt
