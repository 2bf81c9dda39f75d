"""Times the HIP umT5-xxl encoder (24 layers, 512 positions, synthetic weights; 4096-entry vocabulary to keep the
embedding table small).  usage: python tools/t5_bench.py [iters=5] [--cpu]   (one JSON line; --cpu adds the oracle timed on 2 of the 24 layers)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longlive_amd import synth  # noqa: E402
from longlive_amd.text_encoder import WanTextEncoder  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    iters = int(argv[0]) if argv else 5
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
    if "--cpu" in sys.argv:
        from oracle import ref_t5 as RT
        torch.set_num_threads(min(os.cpu_count() or 1, 16))
        sd = {k: v.cpu() for k, v in enc.text_encoder.state_dict().items() if not k.startswith("blocks.") or int(k.split(".")[1]) < 2}
        t0 = time.perf_counter()
        RT.text_encoder_forward(ids, mask, sd, 2, cfg.num_heads)
        cdt = (time.perf_counter() - t0) * cfg.num_layers / 2
        rec["cpu_baseline"] = {"value": 1e3 * cdt, "unit": "ms per prompt", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": "oracle on 2 of 24 layers, extrapolated x12"}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
