#!/usr/bin/env python3
"""ll_rms_silu_cl outputs on seeded inputs (C = 96 / 192 / 384, with and without SiLU, incl. a pixel of zeros and tiny / huge rows) to a
.pt file: run once per library build (LONGLIVE_HIP_LIB=...) and compare with tools/rowkernel_dump.py --compare."""
import sys

import torch

sys.path.insert(0, ".")
from longlive_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(7)
out = {}
for C in (96, 192, 384):
    x = (torch.randn(4099, C, generator=g) * 1.7).to(torch.bfloat16)
    x[5] = 0
    x[6] *= 1e-30
    x[7] *= 1e30
    x[8, 1:] *= 1e-6
    gam = (torch.randn(C, generator=g) * 0.3 + 1).to(torch.bfloat16)
    for silu in (True, False):
        out[f"C{C}.silu{int(silu)}"] = ops.rms_silu_cl(x.cuda(), gam.cuda(), silu=silu).cpu()
torch.save(out, sys.argv[1])
print(f"wrote {len(out)} tensors to {sys.argv[1]}")
