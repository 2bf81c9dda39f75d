set -u
O=gpurun_out/r03ak; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py -m gpu -x -q -k "small_m or t5 or T5 or gemm_asm" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests.log | head -20; exit $rc; }
timeout -k 10 200 python tools/t5_bench.py 5 | tee $O/t5_bench.json
LL_TUNING=gemm_asm=0 timeout -k 10 200 python tools/t5_bench.py 5 | tee -a $O/t5_bench.json
