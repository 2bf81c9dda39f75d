"""Command-line parity with the reference's entry points:

    python -m longlive_amd.cli inference   --config_path configs/longlive_inference.yaml              (inference.py)
    python -m longlive_amd.cli interactive --config_path configs/longlive_interactive_inference.yaml  (interactive_inference.py)

Same yaml keys (configs/longlive_inference.yaml, configs/longlive_interactive_inference.yaml), same prompt file formats
(utils/dataset.py:20-42 one prompt per line; :80-123 jsonl with a "prompts" list per line), same rank partition
(DistributedSampler(shuffle=False, drop_last=True), seed + local_rank: inference.py:49,146), same `inference_iter` stop rule
(inference.py:246) and output names (`rank{r}-{idx}-{seed_idx}_{lora|ema|regular}.mp4`, inference.py:226-243).  Everything
here is host logic; the models are longlive_amd's HIP modules (generator, VAE decoder, umT5 encoder).

Extra key, because this repository cannot ship checkpoints: `synthetic: true` (or `--synthetic`) runs random-init weights
(longlive_amd.synth) and a hash tokenizer, optionally shrunk by `synthetic_overrides: {num_layers, t5_layers, lat_h, lat_w}`.
"""
from __future__ import annotations

import argparse
import json
import os
import struct
import sys
import time
import zlib
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import torch

from .replicas import replica_seed, shard_prompts


# ---- config ------------------------------------------------------------------------------------------------------------
class Config(SimpleNamespace):
    """Attribute view of the yaml (what the reference gets from OmegaConf.load, inference.py:26)."""

    def get(self, key, default=None):
        return getattr(self, key, default)

    def __contains__(self, key):
        return hasattr(self, key)


def _wrap(obj):
    if isinstance(obj, dict):
        return Config(**{k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    return obj


def load_config(path: str) -> Config:
    import yaml
    with open(path, encoding="utf-8") as f:
        raw = yaml.safe_load(f) or {}
    if not isinstance(raw, dict):
        raise ValueError(f"{path}: top level of the config must be a mapping")
    return _wrap(raw)


def parse_switch_frame_indices(value) -> List[int]:
    """interactive_inference.py:146-152: an int, or a comma-separated string ("40, 80, 120")."""
    if isinstance(value, bool):
        raise ValueError("switch_frame_indices must be an int or a comma-separated list")
    if isinstance(value, int):
        return [int(value)]
    if isinstance(value, (list, tuple)):
        return [int(v) for v in value]
    return [int(x) for x in str(value).split(",") if str(x).strip()]


# ---- prompt files ------------------------------------------------------------------------------------------------------
class TextDataset:
    """utils/dataset.py:20-42: one prompt per line (`line.rstrip()`), optional parallel file of extended prompts."""

    def __init__(self, prompt_path: str, extended_prompt_path: Optional[str] = None):
        with open(prompt_path, encoding="utf-8") as f:
            self.prompt_list = [line.rstrip() for line in f]
        self.extended_prompt_list = None
        if extended_prompt_path is not None:
            with open(extended_prompt_path, encoding="utf-8") as f:
                self.extended_prompt_list = [line.rstrip() for line in f]
            assert len(self.extended_prompt_list) == len(self.prompt_list)

    def __len__(self):
        return len(self.prompt_list)

    def __getitem__(self, idx):
        batch = {"prompts": self.prompt_list[idx], "idx": idx}
        if self.extended_prompt_list is not None:
            batch["extended_prompts"] = self.extended_prompt_list[idx]
        return batch


class MultiTextDataset:
    """utils/dataset.py:80-123: jsonl, each line `{"prompts": [segment prompts...]}`, all lines the same length."""

    def __init__(self, prompt_path: str, field: str = "prompts"):
        self.rows: List[List[str]] = []
        with open(prompt_path, encoding="utf-8") as f:
            for i, line in enumerate(f):
                if not line.strip():
                    continue
                ex = json.loads(line)
                assert field in ex, f"Missing field '{field}'"
                val = ex[field]
                assert isinstance(val, list), f"Line {i} field '{field}' is not a list"
                self.rows.append([str(v) for v in val])
        assert len(self.rows) > 0, "JSONL is empty"
        seg_len = len(self.rows[0])
        for i, val in enumerate(self.rows):
            assert len(val) == seg_len, f"Line {i} list length mismatch"
        self.field = field

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx: int):
        return {"idx": idx, "prompts_list": self.rows[idx]}


def rank_indices(n: int, rank: int, world: int) -> List[int]:
    """Sample order of DistributedSampler(dataset, shuffle=False, drop_last=True) on `rank` (inference.py:146), or
    SequentialSampler when world == 1."""
    return shard_prompts(list(range(n)), rank, world) if world > 1 else list(range(n))


def output_name(rank: int, idx: int, seed_idx: int, model_type: str, save_with_index: bool, prompt: str,
                interactive: bool = False, ext: str = ".mp4") -> str:
    """inference.py:236-242 / interactive_inference.py:222-229."""
    if save_with_index:
        return f"rank{rank}-{idx}-{seed_idx}_{model_type}{ext}"
    if interactive:
        return f"rank{rank}-{prompt[:100].replace('/', '_')}-{seed_idx}_{model_type}{ext}"
    return f"rank{rank}-{prompt[:100]}-{seed_idx}{ext}"


def model_type_of(config, lora_enabled: bool) -> str:
    return "lora" if lora_enabled else ("ema" if config.get("use_ema", False) else "regular")


# ---- video files -------------------------------------------------------------------------------------------------------
def write_avi_rgb24(path: str, frames: torch.Tensor, fps: int = 16) -> None:
    """Uncompressed AVI (RIFF, one 'vids' stream of bottom-up BGR24 DIBs + idx1).  frames uint8 [T, H, W, 3] RGB.
    Streamed to disk one frame at a time: all sizes are known up front, so a 3840-frame clip never exists twice in memory."""
    assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3
    T, H, W, _ = frames.shape
    stride = (W * 3 + 3) & ~3
    fsz = stride * H
    avih = struct.pack("<IIIIIIIIII4I", 1000000 // fps, fsz * fps, 0, 0x10, T, 0, 1, fsz, W, H, 0, 0, 0, 0)
    strh = struct.pack("<4s4sIHHIIIIIIII4H", b"vids", b"DIB ", 0, 0, 0, 0, 1, fps, 0, T, fsz, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0, fsz, 0, 0, 0, 0)

    def chunk(tag, payload):
        return tag + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")

    def lst(tag, payload):
        return b"LIST" + struct.pack("<I", len(payload) + 4) + tag + payload

    strl = lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf))
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + strl)
    movi_len = 4 + T * (8 + fsz)                                    # 'movi' + T x ('00db' size data); fsz is even
    idx_len = 16 * T
    riff_len = 4 + len(hdrl) + 8 + movi_len + 8 + idx_len
    pad = bytes(stride - W * 3)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", riff_len) + b"AVI " + hdrl + b"LIST" + struct.pack("<I", movi_len) + b"movi")
        for t in range(T):
            rows = frames[t].flip(0).flip(-1).contiguous().numpy().reshape(H, W * 3)      # bottom-up rows, BGR
            f.write(b"00db" + struct.pack("<I", fsz))
            if pad:
                for r in rows:
                    f.write(r.tobytes() + pad)
            else:
                f.write(rows.tobytes())
        f.write(b"idx1" + struct.pack("<I", idx_len))
        off = 4
        for t in range(T):
            f.write(b"00db" + struct.pack("<III", 0x10, off, fsz))
            off += 8 + fsz


def read_avi_rgb24(path: str) -> torch.Tensor:
    """Inverse of write_avi_rgb24 (tests / round trips)."""
    import numpy as np
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI "
    p = raw.index(b"strf") + 8
    _, W, H = struct.unpack("<Iii", raw[p:p + 12])
    stride = (W * 3 + 3) & ~3
    frames, q = [], raw.index(b"movi") + 4
    while raw[q:q + 4] == b"00db":
        n = struct.unpack("<I", raw[q + 4:q + 8])[0]
        rows = np.frombuffer(raw, dtype=np.uint8, count=n, offset=q + 8).reshape(H, stride)[:, :W * 3].reshape(H, W, 3)
        frames.append(torch.from_numpy(rows[::-1, :, ::-1].copy()))
        q += 8 + n
    return torch.stack(frames, 0)


def write_video(path: str, frames: torch.Tensor, fps: int = 16) -> str:
    """torchvision.io.write_video(path, uint8 [T,H,W,C], fps) (inference.py:243) when torchvision + PyAV exist; otherwise
    the same frames as an uncompressed .avi next to the requested name.  Returns the path written."""
    frames = frames.to(torch.uint8).cpu()
    try:
        from torchvision.io import write_video as tv_write          # noqa: WPS433 (optional dependency)
        import av                                                   # noqa: F401  (what torchvision's writer needs)
    except ImportError:                                             # torchvision / PyAV missing: keep the frames anyway
        alt = os.path.splitext(path)[0] + ".avi"
        write_avi_rgb24(alt, frames, fps)
        return alt
    tv_write(path, frames, fps=fps)                                 # real I/O errors propagate
    return path


# ---- tokenizers --------------------------------------------------------------------------------------------------------
class HashTokenizer:
    """Stand-in for HuggingfaceTokenizer when no tokenizer files exist (synthetic mode): whitespace words -> crc32 ids,
    </s> = 1 appended, <pad> = 0, padded to seq_len.  Same call signature and return types (ids, mask int64)."""

    def __init__(self, vocab_size: int, seq_len: int = 512):
        self.vocab_size, self.seq_len = vocab_size, seq_len

    def __call__(self, texts: Sequence[str], return_mask: bool = True, add_special_tokens: bool = True):
        if isinstance(texts, str):
            texts = [texts]
        ids = torch.zeros(len(texts), self.seq_len, dtype=torch.long)
        mask = torch.zeros_like(ids)
        for b, t in enumerate(texts):
            toks = [2 + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - 2) for w in " ".join(t.split()).split(" ") if w]
            toks = toks[: self.seq_len - 1] + [1]
            ids[b, :len(toks)] = torch.tensor(toks)
            mask[b, :len(toks)] = 1
        return (ids, mask) if return_mask else ids


def hf_tokenizer(path: str, seq_len: int = 512):
    """The reference's HuggingfaceTokenizer(name=path, seq_len=512, clean='whitespace') (wan/modules/tokenizers.py:38-82)
    restated over transformers.AutoTokenizer; ftfy.fix_text is applied when ftfy is installed."""
    import html
    import re
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path)
    try:
        import ftfy
        fix = ftfy.fix_text
    except ImportError:
        fix = lambda s: s                                            # noqa: E731

    def clean(text):
        text = html.unescape(html.unescape(fix(text))).strip()
        return re.sub(r"\s+", " ", text).strip()

    def call(texts, return_mask=True, add_special_tokens=True):
        if isinstance(texts, str):
            texts = [texts]
        out = tok([clean(t) for t in texts], return_tensors="pt", padding="max_length", truncation=True,
                  max_length=seq_len, add_special_tokens=add_special_tokens)
        return (out.input_ids, out.attention_mask) if return_mask else out.input_ids

    return call


# ---- model construction --------------------------------------------------------------------------------------------------
def build_models(config, device, synthetic: bool):
    from . import synth
    from .checkpoint import load_generator
    from .text_encoder import WanTextEncoder
    from .vae import WanVAEWrapper
    from .wan_wrapper import WanDiffusionWrapper
    mk = config.model_kwargs
    ov = config.get("synthetic_overrides", Config()) if synthetic else Config()
    wkw = {k: ov.get(k) for k in ("num_layers", "lat_h", "lat_w") if ov.get(k) is not None}
    wcfg = synth.longlive_1_3b(local_attn_size=mk.local_attn_size, sink_size=mk.sink_size, **wkw)
    lora_enabled = False
    if synthetic:
        tcfg = synth.T5Config(vocab_size=4096, num_layers=ov.get("t5_layers", 24))
        gen = WanDiffusionWrapper(timestep_shift=mk.timestep_shift, local_attn_size=mk.local_attn_size, sink_size=mk.sink_size,
                                  cfg=wcfg, device=device, state_dict=synth.synth_state_dict(wcfg, seed=0, device=device))
        vae = WanVAEWrapper(device=device)
        vae.load_state_dict(synth.synth_vae_state_dict(synth.VaeConfig(), seed=5, device=device))
        enc = WanTextEncoder(tcfg, device=device, tokenizer=HashTokenizer(tcfg.vocab_size, tcfg.text_len))
        enc.load_state_dict(synth.synth_t5_state_dict(tcfg, seed=7, device=device))
    else:
        root = config.get("wan_model_dir", "wan_models/Wan2.1-T2V-1.3B")
        gen = WanDiffusionWrapper(timestep_shift=mk.timestep_shift, local_attn_size=mk.local_attn_size, sink_size=mk.sink_size,
                                  cfg=wcfg, device=device)
        lora = config.get("lora_ckpt") if config.get("adapter") is not None else None
        load_generator(gen, config.generator_ckpt, lora_ckpt=lora, adapter=vars(config.adapter) if lora else None,
                       use_ema=config.get("use_ema", False))
        lora_enabled = lora is not None
        vae = WanVAEWrapper(device=device)
        vae.load_state_dict(torch.load(os.path.join(root, "Wan2.1_VAE.pth"), map_location="cpu"), strict=False)
        enc = WanTextEncoder(device=device, tokenizer=hf_tokenizer(os.path.join(root, "google/umt5-xxl/")))
        enc.load_state_dict(torch.load(os.path.join(root, "models_t5_umt5-xxl-enc-bf16.pth"), map_location="cpu"))
    return wcfg, gen, vae, enc, lora_enabled


def _dist_env():
    if "LOCAL_RANK" in os.environ:
        lr = int(os.environ["LOCAL_RANK"])
        return lr, int(os.environ.get("RANK", str(lr))), int(os.environ.get("WORLD_SIZE", "1"))
    return 0, 0, 1


def run(mode: str, config, synthetic: bool = False, device: Optional[torch.device] = None) -> List[Dict]:
    """The body of inference.py (mode 'inference') / interactive_inference.py (mode 'interactive').  Returns one record per
    written video: {"path", "idx", "frames", "seconds"}."""
    from .pipeline import CausalInferencePipeline, InteractiveCausalInferencePipeline
    local_rank, rank, world = _dist_env()
    if device is None:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    torch.manual_seed(replica_seed(int(config.seed), local_rank))                 # set_seed(config.seed + local_rank)
    synthetic = synthetic or bool(config.get("synthetic", False))
    wcfg, gen, vae, enc, lora_enabled = build_models(config, device, synthetic)
    interactive = mode == "interactive"
    cls = InteractiveCausalInferencePipeline if interactive else CausalInferencePipeline
    pipeline = cls(config, device, generator=gen, text_encoder=enc, vae=vae)
    pipeline.overlap_decode = bool(config.get("overlap_decode", True))     # blocks are decoded while the next ones are generated (same video)
    if interactive:
        switch = parse_switch_frame_indices(config.switch_frame_indices)
        dataset = MultiTextDataset(config.data_path)
        nseg = len(dataset[0]["prompts_list"])
        assert len(switch) == nseg - 1, "The number of switch_frame_indices should be the number of prompt segments minus 1"
    else:
        dataset = TextDataset(prompt_path=config.data_path, extended_prompt_path=config.data_path)
    # every rank creates the folder: the CLI builds no process group, so there is no barrier to order a rank-0-only
    # makedirs before the other ranks' first write (the reference has dist.barrier() there, inference.py:153-156)
    os.makedirs(config.output_folder, exist_ok=True)
    model_type = model_type_of(config, lora_enabled)
    records = []
    for i, idx in enumerate(rank_indices(len(dataset), rank, world)):
        item = dataset[idx]
        noise = torch.randn([config.num_samples, config.num_output_frames, 16, wcfg.lat_h, wcfg.lat_w], device=device,
                            dtype=torch.bfloat16)
        t0 = time.perf_counter()
        if interactive:
            prompts_list = [[p] * config.num_samples for p in item["prompts_list"]]
            video = pipeline.inference(noise=noise, text_prompts_list=prompts_list, switch_frame_indices=switch,
                                       return_latents=False, profile=config.get("profile", False))
            first_prompt = item["prompts_list"][0]
        else:
            prompt = item.get("extended_prompts") or item["prompts"]
            video, _ = pipeline.inference(noise=noise, text_prompts=[prompt] * config.num_samples, return_latents=True,
                                          profile=config.get("profile", False))
            first_prompt = item["prompts"]
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        frames = (255.0 * video.permute(0, 1, 3, 4, 2)).to(torch.uint8).cpu()       # b t c h w -> b t h w c
        vae.model.clear_cache()
        for seed_idx in range(config.num_samples):
            name = output_name(rank, idx, seed_idx, model_type, config.save_with_index, first_prompt, interactive)
            path = write_video(os.path.join(config.output_folder, name), frames[seed_idx], fps=16)
            records.append({"path": path, "idx": idx, "frames": int(frames.shape[1]), "seconds": dt})
        if config.inference_iter != -1 and i >= config.inference_iter:
            break
    return records


def main(argv=None):
    ap = argparse.ArgumentParser(prog="longlive_amd.cli")
    ap.add_argument("mode", choices=["inference", "interactive"])
    ap.add_argument("--config_path", type=str, required=True, help="Path to the config file")
    ap.add_argument("--synthetic", action="store_true", help="random-init weights + hash tokenizer (no checkpoints)")
    args = ap.parse_args(argv)
    recs = run(args.mode, load_config(args.config_path), synthetic=args.synthetic)
    for r in recs:
        print(json.dumps(r))
    return 0


if __name__ == "__main__":
    sys.exit(main())
