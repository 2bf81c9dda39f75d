#!/bin/bash
cd /root/repo
for rep in 1 2; do
  ./tools/kbench shipped 30 | grep -E "gemm_f2|gemm_f1"
  KBENCH_NO_SPLITK=1 ./tools/kbench shipped 30 | grep -E "gemm_f2"
  ./tools/kbench gemmx 30 4680 3072 4480 0 4 | grep custom
done
