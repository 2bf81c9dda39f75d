set -u
O=gpurun_out/r03ae; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "qkv or gemm_asm" > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests.log | head -20; exit $rc; }
for t in "gemm_asm=0" "gemm_asm=3"; do
  echo "== $t"
  LL_TUNING=$t ./tools/kbench gemmx 20 4680 4608 1536 0 2>&1 | grep -E "custom|TFLOP"
done | tee $O/kbench_gemm.txt
for t in 0 3 0 3; do LL_TUNING=gemm_asm=$t ./tools/kbench layerseq 1500 | tee -a $O/kbench_gemm.txt; done
