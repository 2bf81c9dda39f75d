"""CPU checks of the drop-in plumbing that the reference's pipelines rely on: parameter names,
module attributes, and the reference-format cache dicts whose end indices are int64[1] tensors."""
from types import SimpleNamespace

import torch

from longlive_amd import synth
from longlive_amd.model import CausalWanModelHIP, _kv_commit, _kv_state
from longlive_amd.pipeline import CausalInferencePipeline
from longlive_amd.wan_wrapper import WanDiffusionWrapper


def test_state_dict_names_are_the_reference_names():
    cfg = synth.toy_config()
    m = CausalWanModelHIP(cfg, device="cpu")
    want = synth.param_shapes(cfg)     # the same dict loads strictly into the reference's CausalWanModel (make_golden.py)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    sd = synth.synth_state_dict(cfg, seed=3)
    m.load_state_dict(sd, strict=True)
    # wrapper accepts the reference's "model."-prefixed checkpoint keys (inference.py:72-94)
    w = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=3, sink_size=1, cfg=cfg, device="cpu",
                            state_dict={"model." + k: v for k, v in sd.items()})
    assert torch.equal(w.model.blocks[1].ffn[2].weight, sd["blocks.1.ffn.2.weight"])
    assert set(k for k in w.state_dict()) == set("model." + k for k in sd)


def test_max_attention_size_propagation_like_the_reference_pipeline():
    cfg = synth.toy_config(local_attn_size=3, sink_size=1)
    gen = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=3, sink_size=1, cfg=cfg, device="cpu")
    holders = [n for n, mod in gen.model.named_modules() if hasattr(mod, "max_attention_size")]
    assert holders == [""] + [f"blocks.{i}.self_attn" for i in range(cfg.num_layers)]   # root + self-attention only
    args = SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=3, sink_size=1, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=1,
                           context_noise=0)
    P = CausalInferencePipeline(args, "cpu", generator=gen)
    assert (P.num_transformer_blocks, P.frame_seq_length) == (cfg.num_layers, cfg.frame_seqlen)
    P._set_all_modules_max_attention_size(3)
    assert gen.model.max_attention_size == 3 * cfg.frame_seqlen
    assert all(b.self_attn.max_attention_size == 3 * cfg.frame_seqlen for b in gen.model.blocks)
    assert [round(float(t), 2) for t in P.denoising_step_list] == [1000.0, 937.5, 833.33, 625.0]
    assert gen.model._prepare_blockwise_causal_attn_mask(device="cpu", num_frames=3) is None


def test_reference_cache_dicts_with_tensor_indices():
    """pipeline/causal_inference.py:271-277 allocates end indices as int64[1] tensors: they are read once, then only
    written, and the python shadow stays authoritative."""
    cache = {"k": torch.zeros(1, 8, 2, 128), "v": torch.zeros(1, 8, 2, 128),
             "global_end_index": torch.tensor([5]), "local_end_index": torch.tensor([3])}
    assert _kv_state(cache) == (5, 3)
    _kv_commit(cache, 9, 7)
    assert int(cache["global_end_index"]) == 9 and int(cache["local_end_index"]) == 7 and cache["_ll_idx"][:2] == [9, 7]
    assert _kv_state(cache) == (9, 7)             # our own fill_ does not invalidate the shadow (no sync per forward)
    # an EXTERNAL in-place write (the reference's training pipelines: clear_kv_cache() zeroes the index tensors,
    # pipeline/streaming_training.py:290-305) is seen through the tensors' version counters and re-read
    cache["global_end_index"].zero_()
    cache["local_end_index"].zero_()
    assert _kv_state(cache) == (0, 0)
    _kv_commit(cache, 4, 4)
    cache["local_end_index"].fill_(2)
    assert _kv_state(cache) == (4, 2)
    ours = {"global_end_index": 0, "local_end_index": 0}
    _kv_commit(ours, 4, 4)
    assert ours["global_end_index"] == 4 and _kv_state(ours) == (4, 4)


def test_packed_weights_follow_the_parameters():
    """The fused QKV copies must never outlive the parameters they were made from: a load through the WRAPPER
    (inference.py:87/94: `pipeline.generator.load_state_dict`; nn.Module recursion bypasses the child's load_state_dict
    override) and an in-place update (EMA swap, LoRA re-fold) both invalidate them."""
    cfg = synth.toy_config()
    w = WanDiffusionWrapper(timestep_shift=5.0, local_attn_size=3, sink_size=1, cfg=cfg, device="cpu",
                            state_dict=synth.synth_state_dict(cfg, seed=3))
    m = w.model
    old = m._pack()[1]["wqkv"].clone()
    assert m._pack() is m._pack()                                    # cached while nothing changes
    new_sd = synth.synth_state_dict(cfg, seed=4)
    res = w.load_state_dict({"model." + k: v for k, v in new_sd.items()})
    assert not res.missing_keys and not res.unexpected_keys
    got = m._pack()[1]["wqkv"]
    want = torch.cat([new_sd[f"blocks.1.self_attn.{n}.weight"] for n in "qkv"], 0)
    assert torch.equal(got, want) and not torch.equal(got, old)
    # in-place parameter update (autograd's version counter moves)
    with torch.no_grad():
        m.blocks[0].self_attn.k.weight.mul_(2)
    blk0 = m._pack()[0]["wqkv"]
    assert torch.equal(blk0[cfg.dim:2 * cfg.dim], m.blocks[0].self_attn.k.weight)
    # writes through `.data` bypass the version counter: the documented remedy is invalidate_packed()
    m.blocks[0].self_attn.v.weight.data.mul_(3)
    m.invalidate_packed()
    assert torch.equal(m._pack()[0]["wqkv"][2 * cfg.dim:], m.blocks[0].self_attn.v.weight)


def test_uniform_timestep_tag_follows_the_tensor():
    """The pipelines tag their timestep tensors with the host value they were filled with (scheduler.tag_uniform); the tag must
    not survive anything that can change the values: views are untagged objects, an in-place write bumps the version counter."""
    from longlive_amd.scheduler import tag_uniform, uniform_value
    t = tag_uniform(torch.full([2, 3], 750.0), 750.0)
    assert uniform_value(t) == 750.0
    assert uniform_value(t.view(-1)) is None and uniform_value(t.clone()) is None and uniform_value(torch.zeros(3)) is None
    assert uniform_value(t.to("cpu")) == 750.0            # same device: .to() returns the tensor itself
    t[0, 0] = 1.0
    assert uniform_value(t) is None
    pipe = CausalInferencePipeline.__new__(CausalInferencePipeline)
    pipe._timestep_memo = {}
    a, b = pipe._timestep(500.0, 1, 3, "cpu"), pipe._timestep(500.0, 1, 3, "cpu")
    assert a is b and uniform_value(a) == 500.0 and pipe._timestep(500.0, 3, 1, "cpu") is not a


def test_kernel_timer_union_counts_overlapping_launches_once():
    """bench.py's `share_of_step`: launches on two HIP streams overlap; the wall time with at least one of them in flight is the
    union of their event intervals, not the sum of their durations (ops.KernelTimer.union_ms)."""
    from longlive_amd import ops

    class Ev:                                   # a recorded event = a point on one clock
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

    kt = ops.KernelTimer.__new__(ops.KernelTimer)
    kt.tags, kt.base = None, Ev(100.0)
    kt.records = {"a": [(Ev(101.0), Ev(103.0), 1.0), (Ev(110.0), Ev(112.0), 1.0)],        # alone: 2 + 2
                  "b": [(Ev(102.0), Ev(104.5), 1.0), (Ev(111.0), Ev(111.5), 1.0),        # overlaps the first, inside the second
                        (Ev(120.0), Ev(121.0), 1.0)]}
    assert abs(kt.union_ms(("a", "b")) - (3.5 + 2.0 + 1.0)) < 1e-9
    assert abs(kt.union_ms(("a",)) - 4.0) < 1e-9
    assert kt.union_ms(("missing",)) == 0.0
