set -u
O=gpurun_out/r03v; mkdir -p $O
for v in base W_only X_only skel mfma_lds no_epi mem_no_epi skel_no_epi; do
  echo "== $v"
  for shape in "4680 8960 1536 1" "4680 1536 8960 2" "4680 1536 1536 3"; do
    LD_LIBRARY_PATH=experiments/r03/libs/$v LL_TUNING=gemm_asm=1 timeout -k 10 120 ./tools/kbench gemmx 20 $shape 2>&1 | grep -E "custom|TFLOP"
  done
done | tee $O/kbench_gemm_variants.txt
