"""INTEGRATION.md section 1 end to end: the HIP generator driven exactly the way the REFERENCE's pipelines drive their
generator -- cache dicts allocated as pipeline/causal_inference.py:255-293 allocates them (end indices = int64[1] DEVICE
tensors), the call sequence the reference's CausalInferencePipeline / InteractiveCausalInferencePipeline issued
(tests/golden/pipe_calls.pt: timesteps, current_start, recache flag, prompt, and the pipelines' EXTERNAL cache mutations
-- `k.zero_()`, `is_init = False`, interactive_causal_inference.py:39-53,99-103 -- as visible in the recorded cache state
at every call).

Checked after every call: the end indices held in the reference-style tensors, against (i) the same replay on
host-integer caches and (ii) the CPU oracle's state machine; the x0 returned, against the oracle; and at the end the full
cache contents of both cache styles bit for bit."""
import pytest
import torch

from conftest import load_golden
from longlive_amd import synth
from oracle import ref_model as RM
from util import bf, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref_style_caches(n_layers, S):
    # verbatim shapes/dtypes of pipeline/causal_inference.py:271-277,287-292
    kv = [{"k": torch.zeros([1, S, 12, 128], dtype=bf, device=DEV), "v": torch.zeros([1, S, 12, 128], dtype=bf, device=DEV),
           "global_end_index": torch.tensor([0], dtype=torch.long, device=DEV),
           "local_end_index": torch.tensor([0], dtype=torch.long, device=DEV)} for _ in range(n_layers)]
    ca = [{"k": torch.zeros([1, 512, 12, 128], dtype=bf, device=DEV), "v": torch.zeros([1, 512, 12, 128], dtype=bf, device=DEV),
           "is_init": False} for _ in range(n_layers)]
    return kv, ca


def _host_style_caches(n_layers, S):
    kv, ca = _ref_style_caches(n_layers, S)
    for c in kv:
        c["global_end_index"] = 0
        c["local_end_index"] = 0
    return kv, ca


def _idx(c):
    return int(c["global_end_index"]), int(c["local_end_index"])


@pytest.mark.parametrize("tag", ["single", "inter_gs0", "inter_gs1", "train:plain", "train:switch_mid", "train:switch_ext"])
def test_reference_call_log_on_reference_cache_objects(tag):
    """`train:*` = the call logs of the reference's training roll-out pipelines (tests/golden/train_calls.pt): a cache of
    (local_attn_size + 21) frames read through a 12-frame window (pipeline/streaming_training.py:49-50)."""
    from longlive_amd.wan_wrapper import WanDiffusionWrapper
    train = tag.startswith("train:")
    rec = load_golden("train_calls.pt")[tag[6:]] if train else load_golden("pipe_calls.pt")[tag]
    log = rec["log"]
    cfg = synth.WanConfig(num_layers=2, lat_h=4, lat_w=4, local_attn_size=12, sink_size=3)
    fs = cfg.frame_seqlen
    S = (33 if train else 12) * fs
    assert fs == 4 and log[0]["n_layers"] == 2 and log[0]["kv_shape"] == [1, S, 12, 128]
    sd = synth.synth_state_dict(cfg, seed=21)
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=12, sink_size=3, cfg=cfg, device=DEV,
                              state_dict={k: v.to(DEV) for k, v in sd.items()})
    for mod in gen.model.modules():
        if hasattr(mod, "max_attention_size"):
            mod.max_attention_size = 12 * fs
    prompts = {f"p{i}": synth.synth_prompt_embeds(cfg, seed=31 + i) for i in range(4)}
    # oracle (CPU): the reference's arithmetic and state machine restated, pinned bit-exact to the reference's own traces
    om = RM.RefModel(RM.RefConfig.from_cfg(cfg), sd, frame_seqlen_for_max_attn=fs)
    og = RM.RefGenerator(om, 5.0)
    okv = RM.new_kv_cache(1, S, 2, 12, 128)
    oca = RM.new_crossattn_cache(1, 512, 2, 12, 128)

    runs = {"ref": _ref_style_caches(2, S), "host": _host_style_caches(2, S)}
    seq = {"ref": [], "host": [], "oracle": []}
    worst = 0.0
    for i, e in enumerate(log):
        x = synth.hash_normal(77, f"call.{i}", (1, e["frames"], 16, cfg.lat_h, cfg.lat_w)).to(bf)
        t = torch.tensor(e["t"], dtype=torch.float32).view(1, e["frames"])
        x0 = {}
        for style, (kv, ca) in runs.items():
            # the pipeline's external mutations between generator calls, as the recorded cache state shows them
            if e["kv_zero"]:
                for c in kv:
                    c["k"].zero_(); c["v"].zero_()
            if not e["ca_init"]:
                for c in ca:
                    c["k"].zero_(); c["v"].zero_(); c["is_init"] = False
            assert bool(ca[0]["is_init"]) == e["ca_init"]
            _, x0[style] = gen(noisy_image_or_video=x.to(DEV), conditional_dict={"prompt_embeds": prompts[e["prompt"]].to(DEV)},
                               timestep=t.to(DEV), kv_cache=kv, crossattn_cache=ca, current_start=e["cs"],
                               sink_recache_after_switch=e["recache"])
            seq[style].append((_idx(kv[0]), _idx(kv[1])))
            assert all(c["is_init"] for c in ca)
        # the reference keeps its indices in the tensors it allocated: same objects, still int64[1] on the device
        g = runs["ref"][0][0]["global_end_index"]
        assert torch.is_tensor(g) and g.dtype == torch.long and g.is_cuda and g.shape == (1,)
        assert torch.equal(x0["ref"], x0["host"]), f"call {i}: tensor-index and host-index runs differ"
        # oracle
        if e["kv_zero"]:
            for c in okv:
                c["k"].zero_(); c["v"].zero_()
        if not e["ca_init"]:
            for c in oca:
                c["is_init"] = False
        _, r0 = og(x, prompts[e["prompt"]], t, okv, oca, e["cs"], e["recache"])
        seq["oracle"].append(((okv[0]["global_end_index"], okv[0]["local_end_index"]),
                              (okv[1]["global_end_index"], okv[1]["local_end_index"])))
        assert seq["ref"][-1] == seq["oracle"][-1], f"call {i} ({e}): indices {seq['ref'][-1]} vs oracle {seq['oracle'][-1]}"
        r = rel_l2(x0["ref"].cpu(), r0)
        worst = max(worst, r)
        assert r < 2e-2, f"call {i}: x0 relL2 {r}"
    assert seq["ref"] == seq["host"] == seq["oracle"]
    for ca_, cb_, co_ in zip(runs["ref"][0], runs["host"][0], okv):
        assert torch.equal(ca_["k"], cb_["k"]) and torch.equal(ca_["v"], cb_["v"])
        za = ca_["k"].float().abs().sum(dim=(0, 2, 3)).cpu() == 0
        zo = co_["k"].float().abs().sum(dim=(0, 2, 3)) == 0
        assert torch.equal(za, zo), "slot occupancy differs from the oracle"
        assert rel_l2(ca_["k"].cpu(), co_["k"]) < 2e-2 and rel_l2(ca_["v"].cpu(), co_["v"]) < 2e-2
    print(f"{tag}: {len(log)} calls, worst x0 relL2 vs oracle {worst:.2e}, final indices {seq['ref'][-1]}")


class _OracleGen:
    """oracle RefGenerator behind the reference generator's keyword signature (CPU)."""
    supports_kv_only = False

    def __init__(self, og, model_ns):
        self.og, self.model = og, model_ns

    def __call__(self, noisy_image_or_video, conditional_dict, timestep, kv_cache, crossattn_cache, current_start, **kw):
        return self.og(noisy_image_or_video, conditional_dict["prompt_embeds"], timestep.float(), kv_cache, crossattn_cache,
                       current_start, False)


def test_training_rollout_pipeline_on_the_hip_generator():
    """StreamingSwitchTrainingPipeline (forward-only mirror of pipeline/streaming_switch_training.py) end to end on the HIP
    generator: two chunks on one persistent cache, the second with a mid-chunk prompt switch recached from the previous
    chunk's frames; against the same pipeline class driving the CPU oracle."""
    from types import SimpleNamespace
    import trace_driver as TD
    from longlive_amd.pipeline import StreamingSwitchTrainingPipeline
    from longlive_amd.wan_wrapper import WanDiffusionWrapper
    from oracle import ref_ops as R
    cfg = synth.WanConfig(num_layers=2, lat_h=4, lat_w=4, local_attn_size=12, sink_size=3)
    fs = cfg.frame_seqlen
    sd = synth.synth_state_dict(cfg, seed=21)
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=12, sink_size=3, cfg=cfg, device=DEV,
                              state_dict={k: v.to(DEV) for k, v in sd.items()})
    om = RM.RefModel(RM.RefConfig.from_cfg(cfg), sd, frame_seqlen_for_max_attn=fs)
    ons = SimpleNamespace(local_attn_size=12, max_attention_size=0, block_mask=None, named_modules=lambda: [],
                          _prepare_blockwise_causal_attn_mask=lambda **kw: None)
    ogen = _OracleGen(RM.RefGenerator(om, 5.0), ons)
    prompts = [synth.synth_prompt_embeds(cfg, seed=31 + i) for i in range(2)]
    outs = {}
    for name, g, dev in (("hip", gen, DEV), ("oracle", ogen, "cpu")):
        P = StreamingSwitchTrainingPipeline(denoising_step_list=[1000, 750, 500, 250],
                                            scheduler=g.get_scheduler() if name == "hip" else R.FlowMatchSchedulerRef(5.0), generator=g, num_frame_per_block=3, local_attn_size=12, slice_last_frames=21)
        P.num_transformer_blocks, P.frame_seq_length, P.kv_cache_size = 2, fs, 33 * fs
        if name == "hip":
            assert (P.num_heads, P.head_dim, P.text_len) == (12, 128, 512)
            P._initialize_kv_cache(1, bf, dev)
            P._initialize_crossattn_cache(1, bf, dev)
        else:
            P.kv_cache1 = RM.new_kv_cache(1, 33 * fs, 2, 12, 128)
            P.crossattn_cache = RM.new_crossattn_cache(1, 512, 2, 12, 128)
        P.randn_like = TD.HashRandn(5)
        torch.manual_seed(11)
        P.randint = lambda low, high, size, device: torch.randint(low=low, high=high, size=size)     # CPU draw for both
        c = [{"prompt_embeds": p.to(dev)} for p in prompts]
        a, *_ = P.generate_chunk_with_cache(synth.synth_noise(cfg, 6, seed=3).to(dev), c[0], current_start_frame=0, requires_grad=False)
        b, *_ = P.generate_chunk_with_cache(synth.synth_noise(cfg, 9, seed=4).to(dev), c[0], current_start_frame=6, requires_grad=False,
                                            switch_frame_index=3, switch_conditional_dict=c[1], switch_recache_frames=a)
        outs[name] = (a.cpu(), b.cpu(), [(int(k["global_end_index"]), int(k["local_end_index"])) for k in P.kv_cache1],
                      P.kv_cache1[0]["k"].float().cpu())
        if name == "hip":
            assert g.model.max_attention_size == 12 * fs
    assert outs["hip"][2] == outs["oracle"][2]
    for i in (0, 1, 3):
        assert rel_l2(outs["hip"][i], outs["oracle"][i]) < 2e-2, i
