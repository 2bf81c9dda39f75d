set -u
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_asm" > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests.log | head -20; exit $rc; }
for t in "gemm_asm=0" "gemm_asm=1"; do
  echo "== $t"
  for shape in "4680 8960 1536 1" "4680 1536 8960 2" "4680 1536 1536 2" "4680 1536 1536 3" "4680 1536 1536 0"; do
    LL_TUNING=$t ./tools/kbench gemmx 20 $shape 2>&1 | grep -E "custom|TFLOP"
  done
done | tee $O/kbench_gemm.txt
LL_TUNING=gemm_asm=0 ./tools/kbench layerseq 300 | tee -a $O/kbench_gemm.txt
LL_TUNING=gemm_asm=1 ./tools/kbench layerseq 300 | tee -a $O/kbench_gemm.txt
LL_TUNING=gemm_asm=3 ./tools/kbench layerseq 300 | tee -a $O/kbench_gemm.txt
