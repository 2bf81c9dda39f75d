"""TEST INFRASTRUCTURE ONLY (like everything under oracle/): a host build of longlive_amd/csrc/synth_hash.h (g++ -O3 -fopenmp) that
evaluates longlive_amd.synth's counter hash ~100x faster than its int64 tensor form on the CPU -- the 24-layer umT5 golden needs
4.6 G synthetic weights (18 min with tensor ops, seconds here).  Bit-identical by construction (the same header the GPU kernel
compiles; tests/test_synth_hash.py compares both against the tensor form).  `install()` plugs it into synth's CPU path; without a
compiler nothing changes."""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = r"""
#include <stdint.h>
#include "synth_hash.h"
extern "C" void fill(float* out, unsigned long long lo, long long n, unsigned long long stream, int kind) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < n; ++i)
    out[i] = kind ? ll_synth_normal((uint64_t)lo + (uint64_t)i, stream) : ll_synth_uniform((uint64_t)lo + (uint64_t)i, stream);
}
"""
_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    out_dir = os.path.join(_HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so, src = os.path.join(out_dir, "libsynthhash.so"), os.path.join(out_dir, "synthhash.cc")
    hdr = os.path.join(os.path.dirname(_HERE), "longlive_amd", "csrc")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(hdr, "synth_hash.h")):
        open(src, "w").write(_SRC)
        subprocess.run(["g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-I", hdr, src, "-o", so], check=True)
    _lib = C.CDLL(so)
    _lib.fill.argtypes = [C.c_void_p, C.c_ulonglong, C.c_longlong, C.c_ulonglong, C.c_int]
    _lib.fill.restype = None
    return _lib


def cpu_hash(kind: int, stream_const: int, numel: int) -> torch.Tensor:
    out = torch.empty(numel, dtype=torch.float32)
    _load().fill(out.data_ptr(), 0, numel, stream_const & ((1 << 64) - 1), kind)
    return out


def install() -> bool:
    """synth.hash_uniform / hash_normal on the CPU go through the host build from now on.  False (and no change) without g++."""
    from longlive_amd import synth
    try:
        _load()
    except Exception:
        return False
    synth.CPU_HASH = cpu_hash
    return True
