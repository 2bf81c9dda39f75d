"""Times the HIP VAE decoder at the real geometry (60x104 latents -> 480x832 pixels), synthetic weights.
usage: python tools/vae_bench.py [latent_frames=9] [chunk=2]   (prints one JSON line)
roofline: all ll_conv_cl launches of the timed region (HIP events on the launch stream), algorithmic FLOPs / time against
the 2.5 PFLOP/s bf16 MFMA peak.  `python bench.py --workload vae` prints the same record plus the CPU baseline."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longlive_amd import synth  # noqa: E402
from longlive_amd.vae import WanVAEWrapper  # noqa: E402


def conv_flops(cfg, h, w):
    """Useful MACs*2 per steady-state latent frame (4 output frames), from the layer list."""
    dims, layers = synth.vae_decoder_layout(cfg)
    fl, t, hh, ww = 0.0, 1, h, w
    fl += 2 * t * hh * ww * 16 * 16 + 2 * t * hh * ww * 27 * 16 * dims[0]
    for L in layers:
        if L[0] == "res":
            cin, cout = L[2], L[3]
            fl += 2 * t * hh * ww * 27 * (cin * cout + cout * cout)
            if cin != cout:
                fl += 2 * t * hh * ww * cin * cout
        elif L[0] == "attn":
            c = L[2]
            fl += t * (2 * hh * ww * c * 4 * c + 4 * (hh * ww) ** 2 * c)
        else:
            c = L[2]
            if L[0] == "up3d":
                fl += 2 * t * hh * ww * 3 * c * 2 * c
                t *= 2
            hh, ww = 2 * hh, 2 * ww
            fl += 2 * t * hh * ww * 9 * c * (c // 2)
    fl += 2 * t * hh * ww * 27 * dims[-1] * 3
    return fl


def run(n: int = 9, chunk: int = 2):
    cfg = synth.VaeConfig()
    vae = WanVAEWrapper(cfg, device="cuda", chunk=chunk)
    vae.load_state_dict(synth.synth_vae_state_dict(cfg, seed=5, device="cuda"))
    lat = synth.hash_normal(9, "lat", (1, n, 16, 60, 104), "cuda").to(torch.bfloat16)
    vae.model.clear_cache()
    vae.decode_to_pixel(lat[:, :1], use_cache=True)                  # first frame (no temporal upsampling) + warm-up
    vae.decode_to_pixel(lat[:, 1:1 + chunk], use_cache=True)
    from longlive_amd import ops
    torch.cuda.synchronize()
    ops.timer = ops.KernelTimer(tags=("conv",))
    t0 = time.perf_counter()
    out = vae.decode_to_pixel(lat[:, 1 + chunk:], use_cache=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ksum = ops.timer.summary()["conv"]
    ops.timer = None
    m = n - 1 - chunk
    fl = conv_flops(cfg, 60, 104)
    conv_tf = ksum["work_per_launch"] * ksum["launches"] / (ksum["total_ms"] * 1e-3) / 1e12
    rec = {"latent_frames_timed": m, "chunk": chunk, "ms_per_latent_frame": 1e3 * dt / m,
           "pixel_fps": 4 * m / dt, "useful_tflop_per_latent_frame": fl / 1e12,
           "tflops": fl * m / dt / 1e12, "out_shape": list(out.shape),
           "peak_mem_gb": torch.cuda.max_memory_allocated() / 2**30,
           "roofline": {"bound": "mfma", "kernel": "conv_halo_kernel + conv_cl_kernel (all ll_conv_cl launches of the timed region)",
                        "achieved": conv_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": conv_tf / 2500.0,
                        "launches": ksum["launches"], "share_of_time": ksum["total_ms"] * 1e-3 / dt}}
    return rec, vae, lat


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    rec, _, _ = run(int(argv[0]) if argv else 9, int(argv[1]) if len(argv) > 1 else 2)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
