#!/usr/bin/env python3
"""Runs the generated attention kernel (under the current ASM_* schedule knobs) through the CPU emulator: lint + both completion
models + the rescale path.  Exit code 0 = safe to send to a GPU.  A schedule variant that has not passed this is never launched."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_attn_asm_emu as T
for dma in ("buffer",):
    txt = T.G.generate(dma, "LLB")
    probs = T.G.lint(txt)
    assert not probs, probs[:3]
    for mode in ("lazy", "eager"):
        out, ref, steps, m = T.run_case(txt, mode, 256, 6 * 64 + 20)
        assert np.isfinite(out).all() and np.abs(out - ref).max() < 1.2e-2, (mode, np.abs(out - ref).max())
    out, ref, steps, m = T.run_case(txt, "lazy", 72, 5 * 64, seed=3)
    assert np.isfinite(out).all() and np.abs(out - ref).max() < 1.2e-2
    out, ref, steps, m = T.run_case(txt, "lazy", 256, 7 * 64, seed=5, spikes=[(5, 6 * 64 + 10, 3.0), (9, 3, 3.0), (40, 200, 2.5), (200, 130, 3.0)])
    assert np.isfinite(out).all() and np.abs(out - ref).max() < 2e-2
print("emulator check ok")
