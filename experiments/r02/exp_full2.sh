#!/bin/bash
cd /root/repo
O=gpurun_out/full2; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
