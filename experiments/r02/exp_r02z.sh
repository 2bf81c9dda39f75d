#!/bin/bash
# VAE decode on a second stream beside the next block's generation: parity test, then e2e frames/s with / without
cd /root/repo
O=gpurun_out/r02z; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_vae_gpu.py -q -m gpu -x -k "pipeline_streams" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 500 python tools/run_configs.py 60 --vae --e2e-only > $O/e2e.json 2> $O/e2e.err || { tail -5 $O/e2e.err; exit 1; }
cat $O/e2e.json
