#!/bin/bash
cd /root/repo
O=gpurun_out/cli; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_vae_gpu.py tests/test_cli_gpu.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
