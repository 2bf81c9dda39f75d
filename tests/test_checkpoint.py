"""Checkpoint layouts and LoRA fold on CPU, plus (gpu) the folded model against the oracle
running peft's un-merged LoRA forward."""
import pytest
import torch

from longlive_amd import checkpoint as ck
from longlive_amd import synth
from util import bf, rel_l2

TARGETS = ["self_attn.q", "self_attn.k", "self_attn.v", "self_attn.o", "cross_attn.q", "cross_attn.k", "cross_attn.v",
           "cross_attn.o", "ffn.0", "ffn.2"]


def make_lora(cfg, sd, r=8, seed=77, default_suffix=False):
    lora, raw = {}, {}
    for i in range(cfg.num_layers):
        for t in TARGETS:
            name = f"blocks.{i}.{t}"
            out_f, in_f = sd[name + ".weight"].shape
            A = (synth.hash_normal(seed, name + ".A", (r, in_f)) * 0.12).to(bf)
            B = (synth.hash_normal(seed, name + ".B", (out_f, r)) * 0.12).to(bf)
            lora[name] = (A, B)
            sfx = ".default.weight" if default_suffix else ".weight"
            raw[f"base_model.model.{name}.lora_A{sfx}"] = A
            raw[f"base_model.model.{name}.lora_B{sfx}"] = B
    return lora, raw


def test_checkpoint_layouts(tmp_path):
    cfg = synth.toy_config()
    sd = synth.synth_state_dict(cfg, seed=3)
    wrapped = {"model." + k: v for k, v in sd.items()}
    for layout in ({"generator": wrapped}, {"model": wrapped},
                   {"generator": {}, "generator_ema": {k.replace("model.", "model._fsdp_wrapped_module.", 1): v for k, v in wrapped.items()}}):
        use_ema = "generator_ema" in layout
        path = tmp_path / "g.pt"
        torch.save(layout, path)
        got = ck.strip_model_prefix(ck.extract_generator_state_dict(torch.load(path), use_ema=use_ema))
        assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(ValueError):
        ck.extract_generator_state_dict({"critic": {}})


@pytest.mark.parametrize("default_suffix", [False, True])
def test_lora_fold_matches_unmerged_forward(default_suffix):
    cfg = synth.toy_config()
    sd = synth.synth_state_dict(cfg, seed=3)
    lora, raw = make_lora(cfg, sd, r=8, default_suffix=default_suffix)
    folded = ck.fold_lora(sd, {"generator_lora": raw}, rank=8, alpha=16)
    assert set(folded) == set(sd)
    x = synth.hash_normal(5, "x", (7, cfg.dim)).to(bf)
    for name, (A, B) in list(lora.items())[:6]:
        W, b = sd[name + ".weight"], sd[name + ".bias"]
        if W.shape[1] != cfg.dim:
            continue
        want = (x.double() @ W.double().t() + b.double()) + 2.0 * (x.double() @ A.double().t()) @ B.double().t()
        got = torch.nn.functional.linear(x, folded[name + ".weight"], b)
        assert rel_l2(got, want) < 1e-2
        assert not torch.equal(folded[name + ".weight"], W)
    # untouched tensors are passed through
    assert torch.equal(folded["head.head.weight"], sd["head.head.weight"])
    with pytest.raises(ValueError, match="rank"):
        ck.fold_lora(sd, raw, rank=4)
    bad = dict(raw); bad.pop(next(k for k in bad if "lora_B" in k))
    with pytest.raises(ValueError, match="incomplete"):
        ck.fold_lora(sd, bad)
    with pytest.raises(ValueError, match="unrecognised"):
        ck.fold_lora(sd, {"foo": torch.zeros(1)})


@pytest.mark.gpu
def test_folded_lora_model_vs_oracle_unmerged():
    from longlive_amd.wan_wrapper import WanDiffusionWrapper
    from oracle import ref_model as RM
    DEV = "cuda"
    cfg = synth.toy_config(local_attn_size=3, sink_size=1)
    sd = synth.synth_state_dict(cfg, seed=3)
    lora, raw = make_lora(cfg, sd, r=8)
    fs, S = cfg.frame_seqlen, 3 * cfg.frame_seqlen
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=3, sink_size=1, cfg=cfg, device=DEV)
    ck.load_generator(gen, {"generator": {"model." + k: v for k, v in sd.items()}}, {"generator_lora": raw},
                      adapter={"rank": 8, "alpha": 8})
    for m in gen.model.modules():
        if hasattr(m, "max_attention_size"):
            m.max_attention_size = S
    om = RM.RefModel(RM.RefConfig.from_cfg(cfg), sd, frame_seqlen_for_max_attn=fs, lora=lora, lora_scaling=1.0)
    base = RM.RefModel(RM.RefConfig.from_cfg(cfg), sd, frame_seqlen_for_max_attn=fs)
    og, bg = RM.RefGenerator(om, 5.0), RM.RefGenerator(base, 5.0)
    kv = [dict(k=torch.zeros(1, S, cfg.num_heads, 128, dtype=bf, device=DEV), v=torch.zeros(1, S, cfg.num_heads, 128, dtype=bf, device=DEV),
               global_end_index=0, local_end_index=0) for _ in range(cfg.num_layers)]
    ca = [dict(k=torch.zeros(1, cfg.text_len, cfg.num_heads, 128, dtype=bf, device=DEV),
               v=torch.zeros(1, cfg.text_len, cfg.num_heads, 128, dtype=bf, device=DEV), is_init=False) for _ in range(cfg.num_layers)]
    okv, oca = RM.new_kv_cache(1, S, cfg.num_layers, cfg.num_heads, 128), RM.new_crossattn_cache(1, cfg.text_len, cfg.num_layers, cfg.num_heads, 128)
    bkv, bca = RM.new_kv_cache(1, S, cfg.num_layers, cfg.num_heads, 128), RM.new_crossattn_cache(1, cfg.text_len, cfg.num_layers, cfg.num_heads, 128)
    noise = synth.synth_noise(cfg, 2, seed=5)
    prompt = synth.synth_prompt_embeds(cfg, seed=7, valid_tokens=9)
    for f in range(2):
        t = torch.full((1, 1), 937.5)
        _, x0 = gen(noise[:, f:f + 1].to(DEV), {"prompt_embeds": prompt.to(DEV)}, t.to(DEV), kv_cache=kv, crossattn_cache=ca, current_start=f * fs)
        _, r0 = og(noise[:, f:f + 1], prompt, t, okv, oca, f * fs)
        _, b0 = bg(noise[:, f:f + 1], prompt, t, bkv, bca, f * fs)
        assert rel_l2(x0.cpu(), r0) < 2e-2, rel_l2(x0.cpu(), r0)
        assert rel_l2(b0, r0) > 4 * rel_l2(x0.cpu(), r0), "the adapters must matter in this test"
