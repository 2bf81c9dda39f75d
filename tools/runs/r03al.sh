set -u
O=gpurun_out/r03al; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_t5_gpu.py -m gpu -x -q -k "small_m or t5 or T5 or gemm_asm or t5norm" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests.log | head -20; exit $rc; }
timeout -k 10 200 python tools/t5_bench.py 5 | tee $O/t5_bench.json
LL_TUNING=gemm_asm=3 timeout -k 10 60 ./tools/kbench gemmx 20 512 20480 4096 0 2>&1 | grep custom
