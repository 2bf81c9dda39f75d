#!/bin/bash
# tools/build_gemm_variant.sh NAME VAR=1 ... : timing-only build of the generated GEMM kernels (ASM_G_NO_* switches: results INVALID)
# into experiments/r03/libs/NAME/liblonglive_hip.so (run a tool against it with LD_LIBRARY_PATH=experiments/r03/libs/NAME).
# The in-tree library is rebuilt WITHOUT the switches before the script returns, whatever happens in between.
set -e
cd /root/repo
name=$1; shift
clean() {
  rm -f longlive_amd/csrc/build/gemm_asm_*.inc longlive_amd/csrc/build/gemm_asm.o
}
restore() {
  clean
  make -C longlive_amd/csrc -j6 > /dev/null 2>&1 || echo "WARNING: could not rebuild the in-tree library"      # (the Makefile runs the generators in an empty environment)
}
trap restore EXIT
clean
make -C longlive_amd/csrc -j6 GENENV="$*" GENFLAGS=--diag 2>&1 | grep -E " error|lint findings" | grep -v " 0 lint" || true
mkdir -p experiments/r03/libs/$name
cp longlive_amd/liblonglive_hip.so experiments/r03/libs/$name/liblonglive_hip.so
