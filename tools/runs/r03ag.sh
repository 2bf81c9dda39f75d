set -u
O=gpurun_out/r03ag; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_boundary_gpu.py -m gpu -x -q -k "attn or cross or pipeline or block or forward or model or boundary" > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests.log | head -20; exit $rc; }
for t in 1024 512 1024 512; do LL_TUNING=attn_asm_min_keys=$t ./tools/kbench layerseq 1500 | tee -a $O/layerseq.txt; done
