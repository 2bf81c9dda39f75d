set -u
O=gpurun_out/r03ad; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_asm or gate or gemm_bf16" > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests.log | head -20; exit $rc; }
for t in "gemm_asm=0" "gemm_asm=3"; do
  echo "== $t"
  for shape in "4680 8960 1536 1" "4680 1536 8960 2" "4680 1536 1536 2" "4680 1536 1536 3" "4680 1536 1536 0"; do
    LL_TUNING=$t ./tools/kbench gemmx 20 $shape 2>&1 | grep -E "custom|TFLOP"
  done
done | tee $O/kbench_gemm.txt
for t in 0 3 0 3; do LL_TUNING=gemm_asm=$t ./tools/kbench layerseq 1500 | tee -a $O/kbench_gemm.txt; done
