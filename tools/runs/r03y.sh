set -u
O=gpurun_out/r03y; mkdir -p $O
for t in 0 1 0 5 9 3 0; do
  LL_TUNING=gemm_asm=$t python3 tools/pwr_sample.py gemm_asm=$t -- ./tools/kbench layerseq 1500
done 2>&1 | tee $O/layerseq_power.txt
