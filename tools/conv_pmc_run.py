"""One decoder convolution at the real size, a few launches (for rocprofv3 --pmc passes, tools/conv_pmc.sh):
96 -> 96 channels, 3x3x3, 8 frames of 480 x 832 (the launch that is 70 % of the decoder's FLOPs).  Prints its own time."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longlive_amd import ops, synth  # noqa: E402

T, H, W, C = 8, 480, 832, 96
dev = torch.device("cuda", 0)
x = synth.hash_normal(3, "pmc.x", (T + 2, H, W, C)).to(torch.bfloat16).to(dev)
w = (synth.hash_normal(3, "pmc.w", (C, C, 3, 3, 3)) / math.sqrt(27 * C)).to(torch.bfloat16).to(dev)
pk, pb, geo = ops.pack_conv_weight(w, torch.zeros(C, dtype=torch.bfloat16, device=dev))
out = torch.empty(T, H, W, C, dtype=torch.bfloat16, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(2):
    ops.conv_cl(x, pk, pb, geo, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ops.conv_cl(x, pk, pb, geo, out=out)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / n * 1e6
flop = 2.0 * T * H * W * C * C * 27
print(f'#WORK {{"tag": "conv_halo_96", "us": {us:.1f}, "flop": {flop:.0f}, "bytes": {2.0 * ((T + 2) * H * W * C + T * H * W * C + 27 * C * C):.0f}}}')
