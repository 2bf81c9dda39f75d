"""csrc/synth_hash.h (the integer hash the library's ll_synth_hash kernel evaluates on the GPU) against longlive_amd/synth.py's
int64 tensor evaluation: compiled for the HOST with g++ here, so the comparison runs without a GPU; the device side of the same
header is compared in tests/test_ops_gpu.py::test_synth_hash_kernel_matches_the_tensor_hash."""
import os
import subprocess

import numpy as np
import torch

from longlive_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <stdio.h>
#include <stdlib.h>
#include "synth_hash.h"
int main(int argc, char** argv) {
  unsigned long long s = strtoull(argv[1], 0, 10), lo = strtoull(argv[2], 0, 10), n = strtoull(argv[3], 0, 10);
  int kind = atoi(argv[4]);
  for (unsigned long long i = 0; i < n; ++i) {
    float v = kind ? ll_synth_normal(lo + i, s) : ll_synth_uniform(lo + i, s);
    fwrite(&v, 4, 1, stdout);
  }
  return 0;
}
"""


def test_host_build_of_the_hash_header_equals_synth(tmp_path):
    src = tmp_path / "h.cc"
    src.write_text(SRC)
    exe = tmp_path / "h"
    subprocess.run(["g++", "-O1", "-I", os.path.join(ROOT, "longlive_amd", "csrc"), str(src), "-o", str(exe)], check=True)
    for seed, name, n in ((0, "blocks.0.ffn.0.weight", 5000), (7, "noise", 3001), ((1 << 40) + 3, "x", 257)):
        s = synth._stream(seed, name) & synth._M64
        for kind, fn in ((0, synth.hash_uniform), (1, synth.hash_normal)):
            raw = subprocess.run([str(exe), str(s), "0", str(n), str(kind)], check=True, capture_output=True).stdout
            got = np.frombuffer(raw, dtype=np.float32)
            try:
                synth.FORCE_TORCH_HASH = True               # the int64 TENSOR form, not the accelerated CPU path conftest installs
                ref = fn(seed, name, (n,)).numpy()
            finally:
                synth.FORCE_TORCH_HASH = False
            assert got.shape == ref.shape and (got.view(np.uint32) == ref.view(np.uint32)).all(), (seed, name, kind)


def test_accelerated_cpu_hash_equals_the_tensor_form():
    """oracle/fast_hash.py (what conftest and oracle/make_golden.py install for the CPU path) against the tensor form."""
    from oracle import fast_hash
    if not fast_hash.install():
        import pytest
        pytest.skip("no g++ here")
    for seed, name, shape in ((0, "blocks.0.ffn.0.weight", (301, 77)), (5, "noise", (3, 16, 7, 11))):
        for fn in (synth.hash_uniform, synth.hash_normal):
            fast = fn(seed, name, shape)
            try:
                synth.FORCE_TORCH_HASH = True
                ref = fn(seed, name, shape)
            finally:
                synth.FORCE_TORCH_HASH = False
            assert torch.equal(fast, ref), (seed, name, fn.__name__)
