#!/bin/bash
cd /root/repo
O=gpurun_out/vae2; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_vae_gpu.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do timeout -k 10 200 python tools/vae_bench.py 9 2 >> $O/bench.txt 2>$O/bench.err || { tail -5 $O/bench.err; exit 1; }; done
cut -c1-250 $O/bench.txt
python3 tools/conv_pmc_run.py 20
