mkdir -p gpurun_out/r02o
for rep in 1 2; do
for k in ffn2 qkv ffn1 o; do
for wsv in 0 2; do
LL_TUNING=gemm_ws=$wsv timeout -k 10 60 ./tools/kenergy $k 0 2.5 | sed "s/^/ws=$wsv /" | tee -a gpurun_out/r02o/kenergy.txt
done; done
LL_TUNING=gemm_ws=0 timeout -k 10 60 ./tools/kenergy attn 2 2.5 | tee -a gpurun_out/r02o/kenergy.txt
done
