set -u
O=gpurun_out/r03u; mkdir -p $O
for t in "gemm_asm=0" "gemm_asm=1"; do
  echo "== $t"
  for K in 1472 1536 1600 1664 2048 2112; do
    LL_TUNING=$t timeout -k 10 120 ./tools/kbench gemmx 20 4680 8960 $K 1 2>&1 | grep -E "custom|TFLOP"
  done
  for K in 8896 8960 9024 8192 8256; do
    LL_TUNING=$t timeout -k 10 120 ./tools/kbench gemmx 20 4680 1536 $K 2 2>&1 | grep -E "custom|TFLOP"
  done
done | tee $O/kbench_gemm_kstride.txt
