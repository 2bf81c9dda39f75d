set -u
O=gpurun_out/r03av; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gemm_asm or small_m or qkv" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit $rc
for v in base new base new; do
  echo "== $v"
  for shape in "4680 8960 1536 1" "4680 1536 8960 2" "4680 4608 1536 0" "4680 1536 1536 2" "4680 1536 1536 0"; do
    if [ $v = base ]; then LD_LIBRARY_PATH=experiments/r03/libs/base ./tools/kbench gemmx 20 $shape 2>&1 | grep custom; else ./tools/kbench gemmx 20 $shape 2>&1 | grep custom; fi
  done
done | tee $O/kbench.txt
for v in base new base new; do if [ $v = base ]; then LD_LIBRARY_PATH=experiments/r03/libs/base ./tools/kbench layerseq 1500; else ./tools/kbench layerseq 1500; fi; done | tee -a $O/kbench.txt
