#!/usr/bin/env python3
"""MEASUREMENT ONLY.  Which kernel holds which clock: every load below is ONE kernel at its production shape looped alone for a few
seconds on hash-normal data while the card's clock (hwmon + the eight XCDs' gfxclks), power and power-limit residency are sampled
(bench.Telemetry).  Run once per library (LONGLIVE_HIP_LIB=<variant> for the timing-only builds of tools/build_variant.sh) on ONE
device in one gpurun call; prints one JSON line per load.

    python3 tools/clock_matrix.py [--seconds 2.5] [--loads attn,ffn1,ffn2,qkv,ffn1_hip,ffn2_hip,ffn1_i8_hip,ffn1_i8_gen,row]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

import bench
from longlive_amd import _lib, ops, synth

bf16 = torch.bfloat16
L, LK, C, H, D, FF = 4680, 18720, 1536, 12, 128, 8960


def hn(tag, shape, scale=1.0):
    return (scale * synth.hash_normal(9, tag, shape, device="cuda")).to(bf16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--loads", default="attn,ffn1,ffn2,qkv,ffn1_hip,ffn2_hip,ffn1_i8_hip,ffn1_i8_gen,ffn2_i8_hip,ffn2_i8_gen,row")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    lib = _lib.load()

    def tune(k, v):
        _lib.check(lib.ll_set_tuning(k.encode(), int(v)), "ll_set_tuning")

    x = hn("x", (L, C)); hid = hn("hid", (L, FF), 0.5)
    w1, b1 = hn("w1", (FF, C), 0.03), hn("b1", (FF,), 0.1)
    w2, b2 = hn("w2", (C, FF), 0.03), hn("b2", (C,), 0.1)
    wq, bq = hn("wq", (3 * C, C), 0.03), hn("bq", (3 * C,), 0.1)
    o1 = torch.empty(L, FF, dtype=bf16, device="cuda"); o2 = torch.empty(L, C, dtype=bf16, device="cuda"); o3 = torch.empty(L, 3 * C, dtype=bf16, device="cuda")
    q = hn("q", (1, L, H, D)); kc = hn("kc", (1, LK, H, D)); vc = hn("vc", (1, LK, H, D), 0.5); ao = torch.empty_like(q)
    lnw, lnb = hn("lnw", (C,), 0.1) + 1, hn("lnb", (C,), 0.1)
    xq, sx = ops.quantize_rows(x); hq, shid = ops.quantize_rows(hid)
    w1q, s1 = ops.quantize_rows(w1); w2q, s2 = ops.quantize_rows(w2)
    EG, ER = ops.EPI_BIAS_GELU, ops.EPI_BIAS_RES
    wo, bo = hn("wo", (C, C), 0.03), hn("bo", (C,), 0.1)
    ck, cv = hn("ck", (1, 512, H, D)), hn("cv", (1, 512, H, D), 0.5)
    a2 = ao.view(L, C)
    loads = {
        "attn": (lambda: ops.flash_attn(q, kc, vc, [(0, LK)], out=ao), 35, 4.0 * L * LK * D * H),
        "attn_hip": (lambda: ops.flash_attn(q, kc, vc, [(0, LK)], out=ao), -1, 4.0 * L * LK * D * H),      # attn_asm = 0: the HIP ping-pong kernel
        "cross": (lambda: ops.flash_attn(q, ck, cv, [(0, 512)], out=ao), 35, 4.0 * L * 512 * D * H),
        "o": (lambda: ops.gemm(a2, wo, bo, ER, out=o2, res=x), 35, 2.0 * L * C * C),
        "o_hip": (lambda: ops.gemm(a2, wo, bo, ER, out=o2, res=x), 0, 2.0 * L * C * C),
        "qkv_hip": (lambda: ops.gemm(x, wq, bq, 0, out=o3), 0, 2.0 * L * 3 * C * C),
        "ffn1": (lambda: ops.gemm(x, w1, b1, EG, out=o1), 35, 2.0 * L * FF * C),
        "ffn2": (lambda: ops.gemm(hid, w2, b2, ER, out=o2, res=x), 35, 2.0 * L * FF * C),
        "qkv": (lambda: ops.gemm(x, wq, bq, 0, out=o3), 35, 2.0 * L * 3 * C * C),
        "ffn1_hip": (lambda: ops.gemm(x, w1, b1, EG, out=o1), 0, 2.0 * L * FF * C),
        "ffn2_hip": (lambda: ops.gemm(hid, w2, b2, ER, out=o2, res=x), 0, 2.0 * L * FF * C),
        "ffn1_i8_hip": (lambda: ops.gemm_w8a8(xq, sx, w1q, s1, b1, EG, out=o1), 35, 2.0 * L * FF * C),
        "ffn1_i8_gen": (lambda: ops.gemm_w8a8(xq, sx, w1q, s1, b1, EG, out=o1), 51, 2.0 * L * FF * C),
        "ffn2_i8_hip": (lambda: ops.gemm_w8a8(hq, shid, w2q, s2, b2, ER, out=o2, res=x), 35, 2.0 * L * FF * C),
        "ffn2_i8_gen": (lambda: ops.gemm_w8a8(hq, shid, w2q, s2, b2, ER, out=o2, res=x), 51, 2.0 * L * FF * C),
        "row": (lambda: ops.layernorm_affine(x, lnw, lnb, 1e-6, out=o2), 35, 0.0),
    }
    libname = os.environ.get("LONGLIVE_HIP_LIB", "shipped")
    for name in a.loads.split(","):
        fn, asm, flops = loads[name]
        tune("attn_asm", 0 if asm < 0 else 1)
        tune("gemm_asm", 35 if asm < 0 else asm)
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()                      # ~0.7 s of settling before the sampled window
        while time.perf_counter() - t0 < 0.7:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
        tel = bench.Telemetry(0)
        tel.start()
        n, e0, e1 = 0, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < a.seconds:
            for _ in range(50):
                fn()
            n += 50
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        t = tel.stop(0)
        us = e0.elapsed_time(e1) * 1e3 / n
        g = t.get("gpu_metrics_delta") or {}
        rec = dict(lib=libname, load=name, gemm_asm=asm, us_per_launch=round(us, 2), tflops=round(flops / us * 1e-6, 1) if flops else None,
                   sclk_mhz_avg=t.get("sclk_mhz_avg"), xcd_sclk_mhz_avg=t.get("xcd_sclk_mhz_avg"), power_w_avg=t.get("power_w_avg"),
                   ppt_residency=(g.get("ppt_residency_acc", 0) / max(1, g.get("accumulation_counter", 1))) if g else None,
                   energy_mj_per_launch=round((t.get("power_w_avg") or 0) * us * 1e-3, 2))
        print(json.dumps(rec), flush=True)
    tune("gemm_asm", 35)
    tune("attn_asm", 1)


if __name__ == "__main__":
    with torch.no_grad():
        main()
