#!/usr/bin/env python3
"""Headline benchmark: generated frames/s at 832x480 for LongLive-1.3B frame-level AR inference on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one autoregressive block of BASELINE.json's config 2: 3 latent frames (= 12 pixel frames at 832x480)
produced by 4 denoising DiT forwards (warped schedule 1000/937.5/833.3/625, re-noised in between) plus the
clean-context forward that refreshes the block's K/V -- 5 full 30-layer forwards, frame-sink (3) + short-window (12)
self-attention over the KV cache.  W warm-up blocks run first (default 4: the window is then full, so every timed
block is a steady-state block with Lk = 18720, roll + insert -- the regime of the reference's 20.7 FPS figure and the
most expensive one), then exactly K blocks are timed between barrier + device sync on both sides.

Multi-GPU = independent replicas (the path does not shard): every rank generates its own stream
(own seed / prompt, inference.py:49,146), no data-path collective, no RCCL; value = frames of all replicas / max-over-replicas
time.  Two ways to get N replicas:
  * under torch.distributed.run (the driver's launch): RANK / LOCAL_RANK / WORLD_SIZE from the env; the start/stop barrier
    and the scalar MAX / SUM go through a gloo (CPU) process group -- nothing touches xGMI;
  * plain `python bench.py --gpus N` (WORLD_SIZE unset): this process starts N fresh children BEFORE it touches the GPU
    (HIP_VISIBLE_DEVICES = the r-th entry of the inherited mask, seed + r), synchronises their timed regions over pipes (no process group at all) and prints
    the one JSON line itself, with per-replica frames/s.  A child that fails => non-zero exit, no retry.

The JSON line also carries
  roofline     : the dominant kernel (self-attention), timed live with HIP events on its launch stream over the timed
                 region; achieved = algorithmic FLOPs (4 * Lq * Lk * 128 * heads) / avg launch duration
  kernels      : per-kernel table of one more (untimed) steady-state block with EVERY launch bracketed by HIP events:
                 avg us, algorithmic work, achieved, peak, fraction (MFMA kernels against 2.5 PFLOP/s, row kernels against 8 TB/s)
  extras       : side numbers measured after the timed region (bounded): INT8 frames/s on the same workload, prompt-switch
                 (recache) latency, VAE decode frames/s, umT5 ms per prompt -- none of them is part of `value`
  cpu_baseline : the CPU oracle (a port of the reference's PyTorch path, oracle/) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16
MFMA_I8_DENSE_PEAK_TOPS = 5000.0          # 2x the bf16 rate (v_mfma_i32_16x16x64_i8)
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
PIXEL_FRAMES_PER_LATENT = 4               # VAE temporal stride (wan/configs/wan_t2v_1_3B.py:17)
# Energy view (profiles/r05_clock_matrix.md): algorithmic FLOPs of one steady-state block (5 forwards x 30 layers: self-attention 538.27 G,
# six block linears 390.4 G, cross-attention 14.7 G per layer-forward), the board's floor power (what an HBM-bound row kernel draws)
# and the matrix pipe's own dynamic energy per bf16 FLOP (MFMA-only builds of the generated GEMM text, looped) -- committed figures.
FLOP_PER_BLOCK = 141.5e12
BOARD_FLOOR_W = 619.0
MFMA_PJ_PER_FLOP = 0.43
BASELINE_FPS = None                       # BASELINE.json "published": {} -> no number for this exact metric on MI355X
METRIC = "generated frames/sec (832x480) LongLive-1.3B, frame-sink + short-window attention, 4 denoise steps + clean-context pass"
WORKLOAD = ("LongLive-1.3B 832x480 (latent 16x60x104), 3-frame AR blocks at steady state: Lq=4680, Lk=18720 (sink 3 + window "
            "12 frames), 5 DiT forwards/block, 30 layers, random-init weights")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7, help="timed AR blocks (7 blocks = one 5-second clip)")
    ap.add_argument("--warmup", type=int, default=4, help="untimed AR blocks (4 fill the 12-frame window)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--kernels-only", action="store_true", help="the per-kernel table of one more steady-state block, but none of the side numbers (extras)")
    ap.add_argument("--no-extras", action="store_true", help="skip the kernels table and the side numbers after the timed region")
    ap.add_argument("--cpu-layers", type=int, default=30, help="layers of one steady-state forward timed on the CPU (30 = the whole forward, ~20 s)")
    ap.add_argument("--workload", choices=["dit", "vae", "t5"], default="dit",
                    help="dit (default): the headline metric.  vae / t5: the section-8f rows (VAE decoder, umT5 encoder) with "
                         "their own roofline and cpu_baseline objects; single GPU, not the driver's metric")
    ap.add_argument("--quant", choices=["none", "int8"], default="none",
                    help="int8: W8A8 block linears (BASELINE config 5); the headline metric is the default bf16 path")
    ap.add_argument("--stub-workload", action="store_true",
                    help=argparse.SUPPRESS)      # tests only: replaces the GPU body by a sleep (launcher / rank plumbing on CPU)
    return ap.parse_args(argv)


# ---- CPU baselines (the only place outside tests/ and smoke() that may run the oracle) --------------------------------
def cpu_baseline(num_layers_sample: int, dev=None):
    """Times the CPU oracle on ONE steady-state DiT forward (L = 4680 query tokens, full 18720-slot KV cache, roll + insert) --
    all 30 layers by default (~20 s on the GPU box's 16 host cores) -- and scales to a block (x 5 forwards; x 30 / sample layers when
    fewer layers are asked for).  The synthetic weights and cache contents are hashed on the GPU when one is given (bit-identical to
    the CPU hash, which would take minutes for 1.4 G values) and copied to the host; the timed region is the oracle's forward only."""
    from longlive_amd import synth
    from oracle import ref_model as RM

    gdev = dev if dev is not None else "cpu"
    cfg = synth.longlive_1_3b(num_layers=num_layers_sample)
    sd = {k: v.cpu() for k, v in synth.synth_state_dict(cfg, seed=0, layers=list(range(num_layers_sample)), device=gdev).items()}
    fs = cfg.frame_seqlen
    S = 12 * fs
    m = RM.RefModel(RM.RefConfig.from_cfg(cfg), sd, frame_seqlen_for_max_attn=fs)
    kv = RM.new_kv_cache(1, S, num_layers_sample, 12, 128)
    for i, c in enumerate(kv):
        c["k"] = synth.hash_normal(61, f"kv.{i}.k", (1, S, 12, 128), device=gdev).to(torch.bfloat16).cpu()
        c["v"] = (0.5 * synth.hash_normal(61, f"kv.{i}.v", (1, S, 12, 128), device=gdev)).to(torch.bfloat16).cpu()
        c["global_end_index"] = S
        c["local_end_index"] = S
    ca = RM.new_crossattn_cache(1, 512, num_layers_sample, 12, 128)
    x = synth.synth_noise(cfg, 3, seed=0).permute(0, 2, 1, 3, 4)
    prompt = synth.synth_prompt_embeds(cfg, seed=1)
    t = torch.full((1, 3), 625.0)
    # a 1-GPU box's CPU share is 16 cores; more threads than that only oversubscribes the host
    threads = min(torch.get_num_threads(), int(os.environ.get("LONGLIVE_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    with torch.no_grad():
        m.forward(x, t, prompt, kv, ca, current_start=S)
    dt = time.perf_counter() - t0
    fwd = dt * 30.0 / num_layers_sample            # embeddings/head are <0.1% of a forward
    block = 5.0 * fwd
    what = ("one full steady-state DiT forward (30 layers" if num_layers_sample == 30 else f"{num_layers_sample} of 30 layers of one steady-state DiT forward (")
    scale = "x5 forwards per 12-frame block" if num_layers_sample == 30 else f"x{30 / num_layers_sample:.0f} layers x5 forwards per 12-frame block"
    return dict(value=3 * PIXEL_FRAMES_PER_LATENT / block, unit="frames/s", cores=threads, kind="port",
                sample=f"{what}, L=4680, Lk=18720) timed in {dt:.2f}s on {threads} threads, {scale}")


def cpu_baseline_vae(vae, lat):
    """The oracle (oracle/ref_vae.py) decoding the FIRST latent frame (one 480x832 pixel frame, 3.4 TFLOP) on the host."""
    from longlive_amd import synth
    from oracle import ref_vae as RV
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    _, layers = synth.vae_decoder_layout(vae.model.cfg)
    dec = RV.RefVaeDecoder({k: v.cpu() for k, v in vae.model.state_dict().items()}, layers)
    t0 = time.perf_counter()
    ref = RV.decode_to_pixel(dec, lat[:, :1].cpu(), use_cache=False)
    dt = time.perf_counter() - t0
    return {"value": ref.shape[1] / dt, "unit": "pixel frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle decode of the first latent frame (1 pixel frame at 480x832), %.1f s" % dt}


def cpu_baseline_t5(enc, cfg, ids, mask):
    """The oracle (oracle/ref_t5.py) on 2 of the 24 layers at the real widths, extrapolated x12."""
    from oracle import ref_t5 as RT
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    sd = {k: v.cpu() for k, v in enc.text_encoder.state_dict().items()
          if not k.startswith("blocks.") or int(k.split(".")[1]) < 2}
    t0 = time.perf_counter()
    RT.text_encoder_forward(ids, mask, sd, 2, cfg.num_heads)
    dt = (time.perf_counter() - t0) * cfg.num_layers / 2
    return {"value": 1e3 * dt, "unit": "ms per prompt", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "oracle on 2 of 24 layers, extrapolated x12"}


def side_workload(args):
    """`--workload vae|t5`: tools/vae_bench.py / tools/t5_bench.py (HIP path, roofline) + the CPU baseline leg, which
    lives here because only bench.py may run the oracle outside tests."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.workload == "vae":
        import vae_bench
        rec, vae, lat = vae_bench.run(9, 2)
        if not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_vae(vae, lat)
    else:
        import t5_bench
        rec, enc, cfg, ids, mask = t5_bench.run(5)
        if not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_t5(enc, cfg, ids, mask)
    print(json.dumps(rec), flush=True)


# ---- clock / power telemetry over the timed region ---------------------------------------------------------------------
class Telemetry:
    """Samples the card's hwmon files (power1_input [uW], freq1_input = sclk [Hz]) on a side thread every 20 ms while the timed
    region runs, so that runs on different devices of a pool can be compared (devices differ by 8-12 % at identical code: the
    board runs this pipeline at its package-power limit and the clock it can hold there differs per chip).  The card is the one
    whose PCI address matches the torch device; failing that, the one that drew the most power during the region.  amdsmi (when it
    initialises) adds the deltas of the throttle / power-limit residency accumulators of gpu_metrics.  Everything here is best
    effort: a missing file or library leaves nulls, never fails the run."""

    def __init__(self, device_index=0, period=0.02):
        import glob
        import threading
        self.period, self._stop, self._thr = period, threading.Event(), None
        self.cards = []
        for pw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")):
            d = os.path.dirname(pw)
            card = pw.split("/")[4]
            try:
                pci = os.path.basename(os.path.realpath(os.path.join("/sys/class/drm", card, "device")))
            except OSError:
                pci = None
            self.cards.append(dict(card=card, pci=pci, power=pw, freq=os.path.join(d, "freq1_input"), cap=os.path.join(d, "power1_cap"),
                                   p=[], f=[]))
        self.want_pci = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            self.want_pci = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            pass
        self._smi_handle, self._xcd = None, []
        self._smi0 = self._smi(device_index)

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except Exception:
            return None

    def _smi(self, device_index):
        try:
            import amdsmi
            if not getattr(Telemetry, "_smi_init", False):
                amdsmi.amdsmi_init()
                Telemetry._smi_init = True
            hs = amdsmi.amdsmi_get_processor_handles()
            h = hs[device_index] if device_index < len(hs) else hs[0]
            if self.want_pci:
                for x in hs:
                    try:
                        if amdsmi.amdsmi_get_gpu_device_bdf(x).lower() == self.want_pci.lower():
                            h = x
                    except Exception:
                        pass
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            self._smi_handle = (amdsmi, h)
            return {k: v for k, v in m.items() if isinstance(v, (int, float)) and ("residency" in k or "throttle" in k or "acc" in k)}
        except Exception as exc:
            return {"error": repr(exc)[:120]}

    def _loop(self):
        n = 0
        while not self._stop.is_set():
            for c in self.cards:
                c["p"].append(self._read(c["power"]))
                c["f"].append(self._read(c["freq"]))
            n += 1
            if n % 10 == 0 and self._smi_handle is not None:     # every 200 ms: the eight XCDs' own clocks (gpu_metrics current_gfxclks)
                try:
                    amdsmi, h = self._smi_handle
                    clk = [c for c in amdsmi.amdsmi_get_gpu_metrics_info(h).get("current_gfxclks", [])
                           if isinstance(c, (int, float)) and 0 < c < 60000]
                    if clk:
                        self._xcd.append(clk)
                except Exception:
                    self._smi_handle = None
            self._stop.wait(self.period)

    def start(self):
        import threading
        self._thr = threading.Thread(target=self._loop, daemon=True)
        self._thr.start()

    def stop(self, device_index=0):
        self._stop.set()
        if self._thr is not None:
            self._thr.join()
        out = {"sclk_mhz_avg": None, "sclk_mhz_min": None, "power_w_avg": None, "power_w_max": None, "power_cap_w": None,
               "samples": 0, "source": None}
        best = None
        for c in self.cards:
            p = [x for x in c["p"] if x is not None]
            if not p:
                continue
            c["pavg"] = sum(p) / len(p)
            if self.want_pci and c["pci"] and c["pci"].lower() == self.want_pci.lower():
                best = c
                break
            if best is None or c["pavg"] > best["pavg"]:
                best = c
        if best is not None:
            p = [x * 1e-6 for x in best["p"] if x is not None]
            f = [x * 1e-6 for x in best["f"] if x is not None]
            cap = self._read(best["cap"])
            out.update(power_w_avg=sum(p) / len(p), power_w_max=max(p), samples=len(p), power_cap_w=None if cap is None else cap * 1e-6,
                       source=f"hwmon {best['card']} ({best['pci']}), {1e3 * self.period:.0f} ms period, side thread over the timed region"
                              + ("" if (self.want_pci and best["pci"] and best["pci"].lower() == self.want_pci.lower())
                                 else "; card chosen by highest draw (no PCI match)"))
            if f:
                out.update(sclk_mhz_avg=sum(f) / len(f), sclk_mhz_min=min(f))
        if self._xcd:        # a launch with equal work per CU ends with its slowest XCD: the XCDs of one device hold different clocks
            n = min(len(x) for x in self._xcd)
            out["xcd_sclk_mhz_avg"] = [round(sum(x[k] for x in self._xcd) / len(self._xcd), 1) for k in range(n)]
            out["xcd_samples"] = len(self._xcd)
        s1 = self._smi(device_index)
        if self._smi0 and s1 and "error" not in self._smi0 and "error" not in s1:
            out["gpu_metrics_delta"] = {k: s1[k] - self._smi0[k] for k in s1 if k in self._smi0 and s1[k] != self._smi0[k]}
        elif s1 and "error" in s1:
            out["gpu_metrics_delta"] = s1
        return out


# ---- replica synchronisation -------------------------------------------------------------------------------------------
class NoSync:
    """One replica."""
    world = 1

    def barrier(self):
        pass

    end_barrier = barrier

    def gather(self, frames, elapsed, sclk=None, power=None):
        return [dict(rank=0, frames=frames, elapsed=elapsed, sclk_mhz_avg=sclk, power_w_avg=power)]


class GlooSync:
    """Under torch.distributed.run: barrier and scalar exchange over a gloo (CPU) group -- the data path has no collective
    and the timing plumbing does not need the GPUs' interconnect either."""

    def __init__(self, rank, world):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.rank, self.world = dist, rank, world

    def barrier(self):
        self.dist.barrier()

    end_barrier = barrier

    def gather(self, frames, elapsed, sclk=None, power=None):
        """Per-rank (frames, elapsed) and the rank's own device clock / power over its timed region: devices of one node differ by
        3-8 % at identical code, and a scaling curve must be able to tell that spread from a scaling loss."""
        t = torch.zeros(self.world, 4, dtype=torch.float64)
        t[self.rank] = torch.tensor([frames, elapsed, -1.0 if sclk is None else sclk, -1.0 if power is None else power], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        opt = lambda x: None if x < 0 else float(x)
        return [dict(rank=r, frames=float(t[r, 0]), elapsed=float(t[r, 1]), sclk_mhz_avg=opt(t[r, 2]), power_w_avg=opt(t[r, 3]))
                for r in range(self.world)]

    def close(self):
        self.dist.destroy_process_group()


class PipeSync:
    """Child of `python bench.py --gpus N`: the parent is the rendezvous.  '@READY' up, 'GO' down = the start barrier; the
    stop barrier is the parent's max over the children's elapsed times."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def barrier(self):
        print("@READY", flush=True)
        line = sys.stdin.readline()
        if line.strip() != "GO":
            raise SystemExit(f"replica {self.rank}: launcher went away ({line!r})")

    def end_barrier(self):
        pass                 # the parent takes the max over the replicas' own elapsed times

    def gather(self, frames, elapsed, sclk=None, power=None):
        return [dict(rank=self.rank, frames=frames, elapsed=elapsed, sclk_mhz_avg=sclk, power_w_avg=power)]


# ---- the measured body -------------------------------------------------------------------------------------------------
def _pipe_args():
    return SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                           context_noise=0, global_sink=True)


def _plan_text(fn, *a):
    import ctypes as C
    from longlive_amd import _lib
    buf = C.create_string_buffer(256)
    _lib.check(fn(*a, buf, 256), "plan")
    return buf.value.decode()


def kernel_table(summary, quant, mod_table=True, fuse_qn=True):
    """ops.KernelTimer summary of one steady-state block -> rows of (kernel, launches, avg us, work, achieved, peak, frac).
    Kernel names: the launch plan the library reports for the call (ll_gemm_plan_epi / ll_flash_attn_plan); the two entry points with
    exactly one kernel behind them (ll_gemm_bf16_ssq, ll_flash_attn_qnorm) are timed under their own tags by the model, so the name
    follows from the entry point that WAS called, not from a predicate re-evaluated here."""
    from longlive_amd import _lib
    lib = _lib.load()
    i8 = 1 if quant == "int8" else 0
    L, C, F1, LK = 4680, 1536, 8960, 18720
    gemm_shapes = {"gemm_qkv": (L, 3 * C, C), "gemm_o": (L, C, C), "gemm_cq": (L, C, C), "gemm_cq_ssq": (L, C, C), "gemm_co": (L, C, C),
                   "gemm_f1": (L, F1, C), "gemm_f2": (L, C, F1)}
    what = {"gemm_qkv": "self-attn QKV", "gemm_o": "self-attn O + gate-residual", "gemm_cq": "cross-attn Q", "gemm_cq_ssq": "cross-attn Q + row sums of squares (the q RMSNorm's statistics)",
            "gemm_co": "cross-attn O + residual", "gemm_f1": "FFN1 + GELU", "gemm_f2": "FFN2 + gate-residual",
            "flash_attn_self": "self-attention (sink + window from the KV cache)", "flash_attn_cross": "cross-attention (512 text keys)", "flash_attn_cross_qn": "cross-attention (512 text keys), q RMSNorm in its prologue",
            "ln_modulate": "LayerNorm + modulate (norm1 / norm2 / head)", "layernorm_affine": "norm3",
            "rmsnorm": "cross-attn q RMSNorm", "qk_norm_rope_kv_store": "q/k RMSNorm + RoPE + KV insert",
            "kv_roll": "KV window roll", "quantize_rows": "per-token int8 quantisation", "gemm": "embeddings / head GEMMs"}
    rows = []
    for tag, s in sorted(summary.items(), key=lambda kv: -kv[1]["total_ms"]):
        mfma = tag.startswith("gemm") or tag.startswith("flash_attn")
        if tag in gemm_shapes:
            epi = {"gemm_qkv": 0, "gemm_o": 2, "gemm_cq": 0, "gemm_cq_ssq": 0, "gemm_co": 3, "gemm_f1": 1, "gemm_f2": 2}[tag]      # LL_EPI_* of the call
            # plain: 2 = the fused QKV call (V redirect), 1 = an ordinary call, 0 = a per-batch modulation vector rides along
            # (gate-residual calls when the model runs with use_modulation_table = False: those take the HIP kernels)
            plain = 2 if tag == "gemm_qkv" else (0 if (epi == 2 and not mod_table) else 1)
            if tag == "gemm_cq_ssq":                                # ll_gemm_bf16_ssq has one kernel (gemm_asm.hip: gemm_asm_ssq_launch)
                name = "gemm_asm_128_bias_ssq<bf16> tile 256x128 (4 waves x 64 rows, one wave per SIMD, generated schedule), 228 workgroups"
            else:
                name = _plan_text(lib.ll_gemm_plan_epi, *gemm_shapes[tag], i8, epi, plain)
        elif tag == "flash_attn_self":
            name = _plan_text(lib.ll_flash_attn_plan, L, 12, 1, LK, 0, 1)
        elif tag == "flash_attn_cross":
            name = _plan_text(lib.ll_flash_attn_plan, L, 12, 1, 512, 0, 1)
        elif tag == "flash_attn_cross_qn":                          # ll_flash_attn_qnorm has one kernel (attention_asm.hip)
            name = "flash_attn_asm_qn_kernel (4 waves x 64 rows, one wave per SIMD, generated schedule; WanRMSNorm of q in the Q prologue), 228 workgroups"
        else:
            name = {"ln_modulate": "ln_modulate_kernel", "layernorm_affine": "layernorm_affine_kernel", "rmsnorm": "rmsnorm_kernel",
                    "qk_norm_rope_kv_store": "qk_norm_rope_kv_kernel", "kv_roll": "copy_rows_kernel",
                    "quantize_rows": "quantize_rows_reg_kernel<4 | 18>"}.get(tag, tag)
        secs = s["avg_ms"] * 1e-3
        if mfma:
            peak = MFMA_I8_DENSE_PEAK_TOPS if (i8 and tag in gemm_shapes) else MFMA_BF16_DENSE_PEAK_TFLOPS
            ach = s["work_per_launch"] / secs / 1e12
            unit = "TOP/s" if (i8 and tag in gemm_shapes) else "TFLOP/s"
            bound = "mfma"
        else:
            peak, ach, unit, bound = HBM_PEAK_GBS, s["work_per_launch"] / secs / 1e9, "GB/s", "hbm"
        rows.append({"tag": tag, "what": what.get(tag, tag), "kernel": name, "bound": bound, "launches": s["launches"],
                     "avg_us": 1e3 * s["avg_ms"], "total_ms": s["total_ms"], "work_per_launch": s["work_per_launch"],
                     "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak})
    return rows


def run_extras(gen, pipe_cls, interactive_cls, cfg, dev, budget_s=75.0):
    """Side numbers after the timed region (never part of `value`), each with its own achieved / peak where one applies."""
    from longlive_amd import ops, synth
    t_start = time.perf_counter()
    ex = {}

    def left():
        return budget_s - (time.perf_counter() - t_start)

    def fps_of(quant, blocks=4):
        gen.model.set_quant(quant)
        pipe = pipe_cls(_pipe_args(), dev, generator=gen)
        noise = synth.synth_noise(cfg, 3 * (4 + blocks), seed=0, device=dev)
        st = pipe.stream(noise, {"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1, device=dev)})
        for _ in range(4):
            next(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(blocks):
            next(st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return 12 * blocks / dt, 1e3 * dt / blocks

    try:                                                     # BASELINE config 5's arithmetic on the headline workload
        fps, ms = fps_of("int8")
        ex["int8_w8a8"] = {"value": fps, "unit": "frames/s", "ms_per_step": ms,
                           "workload": "same steady-state blocks, W8A8 block linears (QKV, O, cross-q, cross-o, FFN1, FFN2), bf16 attention"}
    except Exception as exc:
        ex["int8_w8a8"] = {"error": repr(exc)}
    finally:
        gen.model.set_quant(None)

    try:                                                     # config 5's throughput mode: two prompt streams batched through one forward
        pipe = pipe_cls(_pipe_args(), dev, generator=gen)
        blocks = 4
        noise = torch.cat([synth.synth_noise(cfg, 3 * (4 + blocks), seed=s, device=dev) for s in (0, 1)])
        st = pipe.stream(noise, {"prompt_embeds": torch.cat([synth.synth_prompt_embeds(cfg, seed=1 + s, device=dev) for s in (0, 1)])})
        for _ in range(4):
            next(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(blocks):
            next(st)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ex["batch2_fps"] = {"value": 2 * 12 * blocks / dt, "unit": "frames/s (both streams together)", "ms_per_step": 1e3 * dt / blocks,
                            "workload": "two independent prompt streams batched through one forward (B = 2: M = 9360 for every GEMM, weights read "
                                        "once, 456 self-attention workgroups), same steady-state blocks, bf16; bit-identical to the streams run "
                                        "one at a time (tests/test_shipped_sizes_gpu.py)"}
        del st, pipe, noise
    except Exception as exc:
        ex["batch2_fps"] = {"error": repr(exc)}

    def two_streams(quant, blocks=6):                        # config 5's throughput mode: two B = 1 streams on two HIP streams of this process
        from longlive_amd.pipeline import InterleavedStreams
        gen.model.set_quant(quant)
        try:
            pipes = [pipe_cls(_pipe_args(), dev, generator=gen) for _ in (0, 1)]
            runner = InterleavedStreams(pipes, dev)
            noises = [synth.synth_noise(cfg, 3 * (4 + blocks), seed=s, device=dev) for s in (0, 1)]
            prompts = [{"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1 + s, device=dev)} for s in (0, 1)]
            st = runner.stream(noises, prompts)
            for _ in range(4):
                next(st)
            torch.cuda.synchronize()
            tele2 = Telemetry(dev.index or 0)
            tele2.start()
            t0 = time.perf_counter()
            for _ in range(blocks):
                next(st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            t2 = tele2.stop(dev.index or 0) or {}
            del st, runner, pipes, noises
            return {"value": 2 * 12 * blocks / dt, "unit": "frames/s (both streams together)", "ms_per_step": 1e3 * dt / blocks,
                    "sclk_mhz_avg": t2.get("sclk_mhz_avg"), "power_w_avg": t2.get("power_w_avg"),
                    "workload": "two independent B = 1 prompt streams on two HIP streams of one process, launches interleaved block by "
                                "block (pipeline/throughput.py), same steady-state blocks, " + ("W8A8 block linears" if quant else "bf16") +
                                "; each stream bit-identical to its solo run (tests/test_model_gpu.py)"}
        finally:
            gen.model.set_quant(None)

    for key, quant in (("two_stream_fps", None), ("int8_two_stream_fps", "int8")):      # the second one is BASELINE config 5's own mode:
        try:                                                                          # several prompt streams per GPU with INT8 linears
            ex[key] = two_streams(quant)
        except Exception as exc:
            ex[key] = {"error": repr(exc)}

    try:                                                     # config 4: prompt switch = recache of the 12-frame window
        I = interactive_cls(_pipe_args(), dev, generator=gen)
        I.global_sink = False
        T = 24
        noise = synth.synth_noise(cfg, T, seed=0, device=dev)
        prompts = [{"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1 + i, device=dev)} for i in range(2)]
        I._setup(noise, T)
        fs = cfg.frame_seqlen
        for c in I.kv_cache1:                                # state of a stream 24 frames in: window full
            c["k"].normal_(); c["v"].normal_()
            c["global_end_index"], c["local_end_index"] = T * fs, 12 * fs
            c.pop("_ll_idx", None)
        lat = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            I._recache_after_switch(noise, T, prompts[rep % 2])
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - t0))
        ms = min(lat[1:])                                    # MIN of the last 2 of 3 repetitions (the first one warms up); all in `all_ms`
        flop = 30 * 3772.5e9                                 # GEMM 1560.5 + attn 2153.1 + cross 58.9 GFLOP per layer
        ex["prompt_switch_latency"] = {"value": ms, "unit": "ms", "higher_is_better": False, "all_ms": lat,
                                       "what": "InteractiveCausalInferencePipeline._recache_after_switch: 12 frames in ONE forward, L = Lk = 18720, "
                                               "global_sink=false (kv_only: the last layer stops after its K/V insert); MIN of the last 2 of 3 "
                                               "repetitions (all three in all_ms)",
                                       "achieved": flop / (ms * 1e-3) / 1e12, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit_rate": "TFLOP/s",
                                       "frac": flop / (ms * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS}
        del I
    except Exception as exc:
        ex["prompt_switch_latency"] = {"error": repr(exc)}

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    if left() > 20:
        try:
            import vae_bench
            rec, vae, lat = vae_bench.run(7, 2)
            del vae, lat
            rec2, vae, lat = vae_bench.run(7, 2)         # short timed region (4 latent frames): keep the better of two passes
            passes = [rec["pixel_fps"], rec2["pixel_fps"]]
            rec = rec2 if rec2["pixel_fps"] > rec["pixel_fps"] else rec
            ex["vae_decode"] = {"value": rec["pixel_fps"], "unit": "pixel frames/s", "ms_per_latent_frame": rec["ms_per_latent_frame"],
                                "achieved": rec["roofline"]["achieved"], "peak": rec["roofline"]["peak"], "unit_rate": "TFLOP/s",
                                "frac": rec["roofline"]["frac"], "passes_pixel_fps": passes,
                                "what": "streaming Wan-VAE decode 60x104 -> 480x832, all conv launches; BEST OF 2 short passes "
                                        "(4 timed latent frames each; both values in passes_pixel_fps)"}
            del vae, lat
        except Exception as exc:
            ex["vae_decode"] = {"error": repr(exc)}
    if left() > 25:
        try:                                                 # live end-to-end rate: every block decoded while the next is generated
            from longlive_amd.vae import WanVAEWrapper
            vcfg = synth.VaeConfig()
            vae = WanVAEWrapper(vcfg, device=dev, chunk=3)
            vae.load_state_dict(synth.synth_vae_state_dict(vcfg, seed=5, device=dev))
            prompt = {"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1, device=dev)}
            P = pipe_cls(_pipe_args(), dev, generator=gen, text_encoder=lambda text_prompts: prompt, vae=vae)
            nb = 11                                          # 5 blocks fill the window (+ one of margin), 5 timed intervals
            noise = synth.synth_noise(cfg, 3 * nb, seed=0, device=dev)
            res = {}
            for key, overlap in (("serial", False), ("overlap_decode", True)):
                times = []
                for _, px in P.stream_video(noise, ["p0"], overlap_decode=overlap):
                    if overlap:
                        px[0, -1, 0, 0, 0].item()            # wait for THIS block's pixels only (the next block keeps running)
                    else:
                        torch.cuda.synchronize()
                    times.append(time.perf_counter())
                torch.cuda.synchronize()
                steady = [(b - a) * 1e3 for a, b in zip(times[5:-1], times[6:])]
                res[key] = {"fps": 12e3 * len(steady) / sum(steady), "ms_per_block": sum(steady) / len(steady), "blocks_timed": len(steady)}
            ex["e2e_live"] = {"value": res["overlap_decode"]["fps"], "unit": "pixel frames/s", "serial": res["serial"],
                              "overlap_decode": res["overlap_decode"],
                              "what": "CausalInferencePipeline.stream_video at steady state: DiT block + streaming VAE decode of that block to "
                                      "480x832 pixels; value = overlap_decode=True (block i decoded on a second HIP stream while block i+1 is "
                                      "generated, bit-identical pixels), `serial` = decode after each block on one stream; MEAN over the timed blocks"}
            del P, vae, noise
        except Exception as exc:
            ex["e2e_live"] = {"error": repr(exc)}
    if left() > 25:
        try:
            import t5_bench
            rec = t5_bench.run(3)[0]
            ex["umt5_encode"] = {"value": rec["ms_per_prompt"], "unit": "ms per prompt", "higher_is_better": False,
                                 "achieved": rec["tflops"], "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit_rate": "TFLOP/s",
                                 "frac": rec["roofline"]["frac"], "what": "umT5-xxl encoder, 24 layers, 512 positions"}
        except Exception as exc:
            ex["umt5_encode"] = {"error": repr(exc)}
    ex["seconds"] = time.perf_counter() - t_start
    torch.cuda.empty_cache()
    return ex


def run_replica(args, rank, world, local_rank, sync):
    """One replica's timed region.  Returns (record for rank 0 or None)."""
    if args.stub_workload:                                   # tests: launcher / rank plumbing without a GPU
        sync.barrier()
        t0 = time.perf_counter()
        time.sleep(0.05 * (rank + 1))
        sync.end_barrier()
        elapsed = time.perf_counter() - t0
        return dict(frames=args.steps * 3 * PIXEL_FRAMES_PER_LATENT, elapsed=elapsed, roofline=None, kernels=None,
                    extras=None, cpu_baseline=None, visible=os.environ.get("HIP_VISIBLE_DEVICES"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from longlive_amd import _lib, ops, synth
    from longlive_amd.pipeline import CausalInferencePipeline, InteractiveCausalInferencePipeline
    from longlive_amd.wan_wrapper import WanDiffusionWrapper

    _lib.load()                                                        # applies LL_TUNING=key=value,... (kernel A/B only)
    cfg = synth.longlive_1_3b(local_attn_size=12, sink_size=3)
    sd = synth.synth_state_dict(cfg, seed=0, device=dev)               # random-init weights of the 1.3B architecture
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=12, sink_size=3, cfg=cfg, device=dev, state_dict=sd)
    del sd
    quant = None if args.quant == "none" else args.quant
    gen.model.set_quant(quant)
    if os.environ.get("LL_FUSE_QN") == "0":                            # kernel A/B only: cross-attention q RMSNorm as its own launch
        gen.model.fuse_cross_qnorm = False
    if os.environ.get("LL_MODTAB") == "0":                             # kernel A/B only
        gen.model.use_modulation_table = False
    if os.environ.get("LL_FUSE_V") == "0":
        gen.model.fuse_v_insert = False
    if os.environ.get("LL_TAB32") == "0":                              # kernel A/B only: LN + modulate from the bf16 table
        gen.model.use_modulation_f32 = False
    pipe = CausalInferencePipeline(_pipe_args(), dev, generator=gen)
    if os.environ.get("LL_OVERLAP") is not None:                       # kernel A/B only: LL_OVERLAP=0 = the one-stream schedule
        pipe.overlap_context = os.environ["LL_OVERLAP"] == "1"
    if os.environ.get("LL_NOMEMO") == "1":                             # A/B only: untagged timestep tensors = sigma, time embedding and
        pipe._timestep = lambda value, b, f, device: torch.full([b, f], value, dtype=torch.float32, device=device)   # modulation table every forward
    extra_blocks = 0 if (args.no_extras or rank != 0) else 1           # one more steady-state block for the kernels table
    nblocks = args.warmup + args.steps + extra_blocks
    T = 3 * nblocks
    assert T <= 1024, "RoPE frame table has 1024 entries"
    # one independent stream per replica (inference.py:49,146: seed + rank, prompts sharded by rank)
    noise = synth.synth_noise(cfg, T, seed=rank, device=dev)
    prompt = {"prompt_embeds": synth.synth_prompt_embeds(cfg, seed=1 + rank, device=dev)}

    stream = pipe.stream(noise, prompt)
    for _ in range(args.warmup):
        next(stream)
    ktimer = None
    if not args.no_kernel_timer and rank == 0:
        ktimer = ops.KernelTimer(tags=("flash_attn_self", "flash_attn_self_co"))
    tele = Telemetry(local_rank)               # every replica samples ITS device's clock / power (per_replica_* in the record)
    torch.cuda.synchronize()
    sync.barrier()
    torch.cuda.synchronize()
    ops.timer = ktimer
    if tele is not None:
        tele.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        next(stream)
    torch.cuda.synchronize()
    sync.end_barrier()
    elapsed = time.perf_counter() - t0
    ops.timer = None
    telemetry = tele.stop(local_rank) if tele is not None else None
    res = dict(frames=args.steps * 3 * PIXEL_FRAMES_PER_LATENT, elapsed=elapsed, roofline=None, kernels=None, extras=None,
               cpu_baseline=None, overlap_context=bool(pipe.overlap_context), telemetry=telemetry,
               visible=os.environ.get("HIP_VISIBLE_DEVICES"))
    if rank != 0:
        return res
    if ktimer is not None and "flash_attn_self" in ktimer.records:
        summ_t = ktimer.summary()
        alone = summ_t["flash_attn_self"]                               # launches of forwards that run ALONE on the device (3 of 5 per block
        co = summ_t.get("flash_attn_self_co")                           # with the context-pass overlap on; the other 2 co-run on two streams)
        n_all = alone["launches"] + (co["launches"] if co else 0)
        total_ms = alone["total_ms"] + (co["total_ms"] if co else 0.0)
        # Headline = the launches whose HIP-event interval IS the kernel's duration (nothing else on the device).  The event interval of a
        # co-running launch also contains the time its workgroups queued for CUs the other stream's kernel held (a generated kernel owns
        # whole CUs), so it is reported apart (`co_running`, `all_launches_by_events`); that the headline nevertheless describes EVERY
        # launch is what the committed kernel trace of the same command shows: begin -> end over all launches of its window, co-running
        # ones included, equals the `alone` figure (`kernel_trace_avg_us`).
        s = alone
        all_ev = dict(launches=n_all, avg_us=1e3 * total_ms / n_all,
                      frac=alone["work_per_launch"] / (total_ms / n_all * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS)
        achieved = s["work_per_launch"] / (s["avg_ms"] * 1e-3) / 1e12
        busy_ms = ktimer.union_ms(("flash_attn_self", "flash_attn_self_co"))   # wall time with >= 1 self-attention launch in flight
        traffic, src = None, None
        plan_now = _plan_text(_lib.load().ll_flash_attn_plan, 4680, 12, 1, 18720, 0, 1)
        pmc = os.path.join(ROOT, "profiles", "r04_pmc_inpipe.json")   # counters of the SAME kernel in the pipeline's launch order
        if os.path.exists(pmc) and "flash_attn_asm_kernel" in plan_now:
            try:
                row = [k for k in json.load(open(pmc))["kernels"] if "flash_attn_asm_kernel" in k["kernel"] and "Lk=512" not in k["kernel"]][0]
                traffic = row["fabric_bytes_per_launch"]
                src = ("profiles/r04_pmc_inpipe.json: rocprofv3 --pmc passes (separate TCC read / write passes) over tools/kbench layerseq -- "
                       "the layer's 12 launches in model order -- on this kernel and shape (TCC_EA0_RDREQ / _32B / TCC_BUBBLE, "
                       "TCC_EA0_WRREQ / _64B; gfx950 x2 read correction); a committed figure, NOT collected in this run (counters need "
                       "their own profiled passes: tools/gpu_batch.sh pmc_inpipe; MFMA utilisation over bench.py itself: pmc_bench)")
            except Exception:
                traffic = None
        trace_us, trace_src = None, None                       # the kernel trace of the same command, committed (begin -> end of every launch)
        try:
            import glob
            tp = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_kernel_summary.md")))[-1]       # the latest round's committed trace
            for line in open(tp):
                if "flash_attn_asm_kernel" in line and "(self" in line:
                    trace_us = float(line.split("|")[3])
                    trace_src = (f"profiles/{os.path.basename(tp)}: rocprofv3 --kernel-trace of `bench.py --steps 3 --warmup 4` (tools/gpu_batch.sh trace), "
                                 "all self-attention launches of its steady-state window; a committed figure, NOT collected in this run")
                    break
        except Exception:
            trace_us = None
        res["roofline"] = {"bound": "mfma", "kernel": plan_now + "; self-attention Lq=4680, Lk=18720, 12 heads", "achieved": achieved,
                           "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_BF16_DENSE_PEAK_TFLOPS,
                           "traffic": traffic, "traffic_source": src, "launches": s["launches"], "avg_us": 1e3 * s["avg_ms"],
                           "flop_per_launch": s["work_per_launch"],
                           "share_of_step": busy_ms / (1e3 * elapsed),
                           "kernel_trace_avg_us": trace_us, "kernel_trace_source": trace_src,
                           "all_launches_by_events": all_ev,
                           "co_running": None if not co else {"launches": co["launches"], "avg_us": 1e3 * co["avg_ms"],
                                                              "frac": co["work_per_launch"] / (co["avg_ms"] * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS},
                           "note": ("context-pass overlap on (the pipeline's default): per block, the clean-context forward and the next block's "
                                    "first denoising forward run side by side on two HIP streams, so 2 of 5 forwards' launches share the device "
                                    "with another kernel.  `achieved` / `avg_us` / `frac` / `launches` = the launches of the 3 forwards that run alone: "
                                    "their HIP-event interval is the kernel's duration.  The event interval of a co-running launch also contains the "
                                    "time its workgroups queued for CUs held by the other stream's kernel (`co_running`, and `all_launches_by_events` = "
                                    "both sets together: upper bounds, not durations).  The committed kernel trace of the same command -- begin -> end "
                                    "of EVERY self-attention launch of its window, co-running ones included -- gives `kernel_trace_avg_us`, which "
                                    "agrees with the headline: the figure describes all launches.  `share_of_step` = wall time with at least one "
                                    "self-attention launch in flight / elapsed (union of the event intervals, not the sum); the `kernels` table is "
                                    "taken on one stream")
                                   if pipe.overlap_context else
                                   "one HIP stream: `share_of_step` = wall time with a self-attention launch in flight / elapsed"}
    if extra_blocks:
        try:
            pipe.overlap_context = False                               # one stream: per-kernel times without a co-running forward
            ops.timer = ops.KernelTimer()                              # every tagged launch of one more steady-state block
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            next(stream)
            torch.cuda.synchronize()
            blk_ms = 1e3 * (time.perf_counter() - t0)
            summ = ops.timer.summary()
            ops.timer = None
            rows = kernel_table(summ, quant, mod_table=bool(getattr(gen.model, "use_modulation_table", True)),
                                fuse_qn=bool(getattr(gen.model, "fuse_cross_qnorm", True)))
            res["kernels"] = {"note": "one untimed steady-state block, HIP events around every launch (adds ~2 us of gap per launch: "
                                      f"this block took {blk_ms:.1f} ms); shares are of the sum of kernel time",
                              "sum_kernel_ms": sum(r["total_ms"] for r in rows), "rows": rows}
        except Exception as exc:
            ops.timer = None
            res["kernels"] = {"error": repr(exc)}
        del stream, pipe
        if not args.kernels_only:
            try:
                res["extras"] = run_extras(gen, CausalInferencePipeline, InteractiveCausalInferencePipeline, cfg, dev)
            except Exception as exc:
                res["extras"] = {"error": repr(exc)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            del gen                                                    # (the GPU model is no longer needed: the host copy of the weights is the oracle's)
            torch.cuda.empty_cache()
            res["cpu_baseline"] = cpu_baseline(args.cpu_layers, dev)
        except Exception as exc:      # the baseline is reporting only; never lose the GPU number over it
            res["cpu_baseline"] = {"error": repr(exc)}
    return res


def energy_view(elapsed_s: float, steps: int, telemetry) -> dict | None:
    """What a block cost in joules over the timed region, and how far the path is from its POWER roofline: a block of nothing but its
    MFMAs would need FLOP_PER_BLOCK x MFMA_PJ_PER_FLOP of dynamic energy; the board gives (regulated power - floor) of dynamic power."""
    p = (telemetry or {}).get("power_w_avg")
    if not p or steps <= 0:
        return None
    t = elapsed_s / steps
    dyn = (p - BOARD_FLOOR_W) * t
    mfma = FLOP_PER_BLOCK * MFMA_PJ_PER_FLOP * 1e-12
    return {"j_per_block": p * t, "floor_w": BOARD_FLOOR_W, "dynamic_j_per_block": dyn, "mfma_only_j_per_block": mfma,
            "power_roofline_frac": mfma / dyn if dyn > 0 else None,
            "note": "floor_w and the matrix pipe's pJ / FLOP are committed figures of profiles/r05_clock_matrix.md (one device); power is this "
                    "run's; bf16 pipeline only"}


def _clock_power(res) -> dict:
    t = res.get("telemetry") or {}
    return dict(sclk_mhz_avg=t.get("sclk_mhz_avg"), power_w_avg=t.get("power_w_avg"))


def final_record(args, world, per_replica, res0):
    frames = sum(r["frames"] for r in per_replica)
    elapsed = max(r["elapsed"] for r in per_replica)
    fps = frames / elapsed
    out = {
        "metric": METRIC, "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": (fps / BASELINE_FPS) if BASELINE_FPS else None,
        "dtype": "bf16" if args.quant == "none" else "int8 (W8A8 block linears, bf16 attention/norms)", "data": "synthetic",
        "config": {"workload": WORKLOAD, "frames_per_step": 3 * PIXEL_FRAMES_PER_LATENT,
                   "ms_per_latent_frame": 1e3 * elapsed / args.steps / 3, "replicas": world,
                   "parallelism": f"replicas x{world} (no collective on the data path, no RCCL)",
                   "schedule": ("two HIP streams per replica: a block's clean-context forward runs beside the next block's first denoising "
                                "forward (pipeline default, bit-identical to one stream; LL_OVERLAP=0 = one stream)")
                               if res0.get("overlap_context") else "one HIP stream per replica",
                   "per_replica_fps": [r["frames"] / r["elapsed"] for r in sorted(per_replica, key=lambda r: r["rank"])],
                   "per_replica_visible_devices": [r.get("visible") for r in sorted(per_replica, key=lambda r: r["rank"])],
                   "per_replica_sclk_mhz_avg": [r.get("sclk_mhz_avg") for r in sorted(per_replica, key=lambda r: r["rank"])],
                   "per_replica_power_w_avg": [r.get("power_w_avg") for r in sorted(per_replica, key=lambda r: r["rank"])]},
        "roofline": res0.get("roofline"), "cpu_baseline": res0.get("cpu_baseline"),
        "telemetry": res0.get("telemetry"),
        "energy": energy_view(res0["elapsed"], args.steps, res0.get("telemetry")) if (args.quant == "none" and "elapsed" in res0) else None,
    }
    if res0.get("kernels") is not None:
        out["kernels"] = res0["kernels"]
    if res0.get("extras") is not None:
        out["extras"] = res0["extras"]
    return out


# ---- launcher: `python bench.py --gpus N` without torchrun -------------------------------------------------------------
def child_visibility(parent_env, r: int, ndev: int) -> dict:
    """Device mask of replica r, translated THROUGH the parent's own mask: `ndev` was counted under the parent's visibility, so
    replica r must get the r-th entry of the inherited HIP_VISIBLE_DEVICES (or CUDA_VISIBLE_DEVICES) list, not the bare index r --
    with an allotment such as HIP_VISIBLE_DEVICES=4,5 the bare index would put the children on physical GPUs 0 and 1, outside the
    allotted set.  ROCR_VISIBLE_DEVICES is left untouched: HIP indices are relative to it.  Both HIP_ and CUDA_VISIBLE_DEVICES are
    set to the same single entry so that an inherited one cannot compose with the new one."""
    mask = parent_env.get("HIP_VISIBLE_DEVICES")
    if mask is None or mask.strip() == "":
        mask = parent_env.get("CUDA_VISIBLE_DEVICES")
    slot = r % max(1, ndev)
    if mask is not None and mask.strip() != "":
        entries = [e.strip() for e in mask.split(",") if e.strip() != ""]
        if slot >= len(entries):
            raise SystemExit(f"bench.py launcher: replica {r} needs entry {slot} of the inherited device mask {mask!r}")
        dev = entries[slot]
    else:
        dev = str(slot)
    return {"HIP_VISIBLE_DEVICES": dev, "CUDA_VISIBLE_DEVICES": dev}


def launch_replicas(args, argv):
    """Starts N children before this process has touched the GPU (no torch.cuda call above this line in the parent), one
    device each, and acts as their rendezvous.  No retry: any child failure ends the run with a non-zero exit."""
    n = args.gpus
    ndev = n
    if not args.stub_workload:
        ndev = torch.cuda.device_count()                       # counting devices does not initialise HIP on this image
        if ndev < n and os.environ.get("LL_BENCH_SHARE_GPUS") != "1":     # (=1: rehearsal of the launcher on fewer GPUs)
            raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible")
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(LL_BENCH_CHILD="1", LL_BENCH_RANK=str(r), LL_BENCH_WORLD=str(n),
                   HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(child_visibility(os.environ, r, ndev))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, text=True, bufsize=1))

    def fail(msg):
        for p in procs:
            if p.poll() is None:
                p.kill()
        raise SystemExit(f"bench.py launcher: {msg}")

    def read_until(p, r, prefix):
        while True:
            line = p.stdout.readline()
            if line == "":
                fail(f"replica {r} exited (code {p.wait()}) before '{prefix}'")
            if line.startswith(prefix):
                return line[len(prefix):].strip()
            sys.stderr.write(f"[replica {r}] {line}")

    for r, p in enumerate(procs):
        read_until(p, r, "@READY")
    for p in procs:                                               # start barrier: every replica is warmed up and synced
        p.stdin.write("GO\n")
        p.stdin.flush()
    results = [json.loads(read_until(p, r, "@RESULT")) for r, p in enumerate(procs)]
    for r, p in enumerate(procs):
        p.stdin.close()
        if p.wait() != 0:
            fail(f"replica {r} exited with code {p.returncode}")
    per = [dict(rank=r, frames=res["frames"], elapsed=res["elapsed"], visible=res.get("visible"), **_clock_power(res))
           for r, res in enumerate(results)]
    print(json.dumps(final_record(args, n, per, results[0])), flush=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.workload != "dit":
        return side_workload(args)
    if os.environ.get("LL_BENCH_CHILD") == "1":                   # a replica started by launch_replicas
        rank, world = int(os.environ["LL_BENCH_RANK"]), int(os.environ["LL_BENCH_WORLD"])
        res = run_replica(args, rank, world, 0, PipeSync(rank, world))
        print("@RESULT " + json.dumps(res), flush=True)
        return
    if "WORLD_SIZE" in os.environ:                                # torch.distributed.run
        rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
        sync = GlooSync(rank, world) if world > 1 else NoSync()
        res = run_replica(args, rank, world, local_rank, sync)
        cp = _clock_power(res)
        per = sync.gather(res["frames"], res["elapsed"], cp["sclk_mhz_avg"], cp["power_w_avg"])
        if rank == 0:
            print(json.dumps(final_record(args, world, per, res)), flush=True)
        if world > 1:
            sync.close()
        return
    if args.gpus > 1:
        return launch_replicas(args, argv)
    res = run_replica(args, 0, 1, 0, NoSync())
    print(json.dumps(final_record(args, 1, [dict(rank=0, frames=res["frames"], elapsed=res["elapsed"], **_clock_power(res))], res)), flush=True)


if __name__ == "__main__":
    main()
