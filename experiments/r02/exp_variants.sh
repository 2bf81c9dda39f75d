#!/bin/bash
# the opt-in kernel variants under the parity tests (LL_TUNING_TEST applies the key before every test)
cd /root/repo
O=gpurun_out/variants; mkdir -p $O
for v in gemm_ws=2 gemm_stagger=1 gemm_lds_epi=0 gemm_group_m=1 attn_sk_wgs=0 attn_variant=1 gemm_splitk_l2=1; do
  LL_TUNING_TEST=$v timeout -k 10 400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -m gpu -x > $O/$v.log 2>&1
  echo "$v rc=$? $(tail -1 $O/$v.log)"
done
LL_TUNING_TEST=conv_halo=0 timeout -k 10 300 python -m pytest tests/test_vae_gpu.py -q -m gpu -x > $O/conv_halo0.log 2>&1; echo "conv_halo=0 rc=$? $(tail -1 $O/conv_halo0.log)"
