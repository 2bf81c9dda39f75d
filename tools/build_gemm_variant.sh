#!/bin/bash
# tools/build_gemm_variant.sh NAME VAR=1 ... : timing-only build of the generated GEMM kernels into experiments/r03/libs/NAME
set -e
cd /root/repo
name=$1; shift
rm -f /root/repo/longlive_amd/csrc/build/gemm_asm_192_0.inc /root/repo/longlive_amd/csrc/build/gemm_asm_128_0.inc /root/repo/longlive_amd/csrc/build/gemm_asm_128_2.inc /root/repo/longlive_amd/csrc/build/gemm_asm_128_3.inc /root/repo/longlive_amd/csrc/build/gemm_asm_224_1.inc /root/repo/longlive_amd/csrc/build/gemm_asm.o
env "$@" make -C /root/repo/longlive_amd/csrc -j6 2>&1 | grep -E " error|lint findings" | grep -v " 0 lint" || true
mkdir -p /root/repo/experiments/r03/libs/$name
cp /root/repo/longlive_amd/liblonglive_hip.so /root/repo/experiments/r03/libs/$name/liblonglive_hip.so
