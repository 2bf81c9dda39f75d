set -u
O=gpurun_out/r03ah; mkdir -p $O
for v in base mem mfma_lds no_epi skel skel_no_epi; do
  echo "== $v"
  for shape in "4680 8960 1536 1" "4680 1536 8960 2" "4680 4608 1536 0" "4680 1536 1536 2"; do
    LD_LIBRARY_PATH=experiments/r03/libs/$v LL_TUNING=gemm_asm=3 timeout -k 10 120 ./tools/kbench gemmx 20 $shape 2>&1 | grep -E "custom"
  done
done | tee $O/kbench_gemm_v3_variants.txt
