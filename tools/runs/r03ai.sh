set -u
O=gpurun_out/r03ai; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "qkv or gemm_asm or gate" > $O/tests.log 2>&1; rc=$?; tail -4 $O/tests.log; [ $rc -eq 0 ] || { grep -E "^E " $O/tests.log | head -20; exit $rc; }
for shape in "4680 8960 1536 1" "4680 1536 8960 2" "4680 4608 1536 0" "4680 1536 1536 2" "4680 1536 1536 3" "4680 1536 1536 0"; do LL_TUNING=gemm_asm=3 timeout -k 10 60 ./tools/kbench gemmx 20 $shape 2>&1 | grep -E "custom"; done | tee $O/kbench_gemm.txt
LD_LIBRARY_PATH=experiments/r03/libs/base ./tools/kbench layerseq 1500 | tee -a $O/kbench_gemm.txt
./tools/kbench layerseq 1500 | tee -a $O/kbench_gemm.txt
LD_LIBRARY_PATH=experiments/r03/libs/base ./tools/kbench layerseq 1500 | tee -a $O/kbench_gemm.txt
./tools/kbench layerseq 1500 | tee -a $O/kbench_gemm.txt
