set -u
O=gpurun_out/r03z; mkdir -p $O
for k in ffn1 o ffn2; do
  for t in 0 3 0 3; do
    echo "== kenergy $k gemm_asm=$t"
    LL_TUNING=gemm_asm=$t timeout -k 10 60 ./tools/kenergy $k 0 2
  done
done 2>&1 | tee $O/kenergy_gemm.txt
echo "== kenergy ffn2 split-K (shipped), gemm_asm=0"; KENERGY_SPLITK=1 LL_TUNING=gemm_asm=0 timeout -k 10 60 ./tools/kenergy ffn2 0 2 | tee -a $O/kenergy_gemm.txt
