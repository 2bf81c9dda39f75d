"""Times the HIP umT5-xxl encoder (24 layers, 512 positions, synthetic weights; 4096-entry vocabulary to keep the
embedding table small).  usage: python tools/t5_bench.py [iters=5]   (one JSON line; `python bench.py --workload t5` adds the CPU baseline)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longlive_amd import synth  # noqa: E402
from longlive_amd.text_encoder import WanTextEncoder  # noqa: E402


def run(iters: int = 5):
    cfg = synth.T5Config(vocab_size=4096)
    enc = WanTextEncoder(cfg, device="cuda")
    enc.load_state_dict(synth.synth_t5_state_dict(cfg, seed=7, device="cuda"))
    ids, mask = synth.synth_token_ids(cfg, 77, seed=3)
    out = enc.encode_ids(ids, mask)["prompt_embeds"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = enc.encode_ids(ids, mask)["prompt_embeds"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    L, d, f = cfg.text_len, cfg.dim, cfg.dim_ffn
    flops = cfg.num_layers * (2 * L * d * (4 * d + 3 * f) + 4 * L * L * d)
    rec = {"ms_per_prompt": 1e3 * dt, "tflop": flops / 1e12, "tflops": flops / dt / 1e12,
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": 2500.0, "unit": "TFLOP/s", "frac": flops / dt / 2.5e15},
           "finite": bool(torch.isfinite(out.float()).all()), "std_valid_rows": float(out[0, :77].float().std())}
    return rec, enc, cfg, ids, mask


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    print(json.dumps(run(int(argv[0]) if argv else 5)[0]))


if __name__ == "__main__":
    main()
