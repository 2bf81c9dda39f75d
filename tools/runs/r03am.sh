set -u
O=gpurun_out/r03am; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E |^FAILED" $O/tests.log | head -20; exit $rc; }
