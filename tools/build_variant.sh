#!/bin/bash
# tools/build_variant.sh NAME [VAR=VALUE ...] : a second build of the library with generator knobs (ASM_* for the attention kernel,
# ASM_G_* for the GEMM kernels) into experiments/libs/NAME/liblonglive_hip.so, for interleaved A/Bs on one device:
#     LD_LIBRARY_PATH=experiments/libs/NAME ./tools/kbench layerseq 1500      (kbench's RUNPATH comes after LD_LIBRARY_PATH)
#     LONGLIVE_HIP_LIB=experiments/libs/NAME/liblonglive_hip.so python bench.py ...
# Schedule knobs are recorded in the variant's ll_*_plan strings; timing-only knobs (NO_*: results INVALID) switch the generators to
# --diag.  The in-tree library is rebuilt WITHOUT knobs before the script returns, whatever happens in between.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
B=longlive_amd/csrc/build
clean() { rm -f $B/attn_asm_body*.inc $B/gemm_asm*.inc $B/attention_asm.o $B/gemm_asm.o $B/asm_knobs.h $B/api.o; }
restore() { clean; make -C longlive_amd/csrc -j8 > /dev/null 2>&1 || echo "WARNING: could not rebuild the in-tree library"; }
trap restore EXIT
flags=""
case "$*" in *NO_*) flags="--diag" ;; esac
clean
make -C longlive_amd/csrc -j8 GENENV="$*" GENFLAGS="$flags" 2>&1 | grep -E " error|lint findings|refused" | grep -v " 0 lint" || true
mkdir -p experiments/libs/$name
cp longlive_amd/liblonglive_hip.so experiments/libs/$name/liblonglive_hip.so
python3 - <<PY
import ctypes as C
lib = C.CDLL("experiments/libs/$name/liblonglive_hip.so")
b = C.create_string_buffer(400)
lib.ll_flash_attn_plan(4680, 12, 1, 18720, 0, 1, b, 400)
print("variant $name:", b.value.decode()[-120:])
PY
