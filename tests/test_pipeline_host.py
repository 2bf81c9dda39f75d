"""Host orchestration of the two pipelines, on CPU with a fake generator (no model, no GPU): the sequence of generator
calls (timesteps, current_start, recache flags, prompt segment, cache resets) and the resulting latents must equal what
the REFERENCE's CausalInferencePipeline / InteractiveCausalInferencePipeline do with the same fake generator
(tests/golden/pipe_calls.pt, recorded by oracle/make_golden.py::gen_pipe_calls)."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from longlive_amd import synth
from longlive_amd.pipeline import CausalInferencePipeline, InteractiveCausalInferencePipeline
from oracle import ref_ops as R
import trace_driver as TD


class FakeGenerator(nn.Module):
    """Same behaviour as oracle/make_golden.py::FakeGenerator (kept separate: the golden generator is not importable
    on the GPU box, and tests may not depend on /root/reference)."""

    def __init__(self, scheduler, fs):
        super().__init__()
        self.dummy = nn.Parameter(torch.zeros(1))
        self.scheduler = scheduler
        self.log = []
        self.model = SimpleNamespace(num_frame_per_block=1, local_attn_size=-1, max_attention_size=0, block_mask=None,
                                     named_modules=lambda: [], _prepare_blockwise_causal_attn_mask=lambda **kw: None)
        self.fs = fs

    def get_scheduler(self):
        return self.scheduler

    def forward(self, noisy_image_or_video, conditional_dict, timestep, kv_cache=None, crossattn_cache=None,
                current_start=None, sink_recache_after_switch=False, **kw):
        x = noisy_image_or_video
        self.log.append(dict(t=[round(float(v), 4) for v in timestep.flatten().tolist()], cs=int(current_start),
                             recache=bool(sink_recache_after_switch), prompt=conditional_dict["name"],
                             frames=int(x.shape[1]), xsum=round(float(x.float().sum()), 3),
                             kv_zero=bool(kv_cache[0]["k"].abs().sum() == 0),
                             ca_init=bool(crossattn_cache[0]["is_init"]), n_layers=len(kv_cache),
                             kv_shape=list(kv_cache[0]["k"].shape)))
        n = x.shape[1] * self.fs
        kv_cache[0]["k"][:, : min(n, kv_cache[0]["k"].shape[1])] += 1
        crossattn_cache[0]["is_init"] = True
        x0 = (0.5 * x.float() + 0.01 * (len(self.log) % 7)).to(x.dtype)
        return x0, x0


def _args(gs):
    return SimpleNamespace(model_kwargs=SimpleNamespace(local_attn_size=12, sink_size=3, timestep_shift=5.0),
                           denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                           context_noise=0, global_sink=gs)


@pytest.mark.parametrize("tag", ["single", "inter_gs0", "inter_gs1"])
def test_call_sequence_matches_reference_pipeline(tag):
    rec = load_golden("pipe_calls.pt")[tag]
    cfg = synth.WanConfig(lat_h=4, lat_w=4)
    fg = FakeGenerator(R.FlowMatchSchedulerRef(5.0), 4)
    enc = lambda text_prompts: {"prompt_embeds": torch.zeros(1, 1), "name": text_prompts[0]}
    noise = synth.synth_noise(cfg, rec["T"], seed=3)
    if rec["switches"] is None:
        P = CausalInferencePipeline(_args(rec["global_sink"]), "cpu", generator=fg, text_encoder=enc)
    else:
        P = InteractiveCausalInferencePipeline(_args(rec["global_sink"]), "cpu", generator=fg, text_encoder=enc)
    P.num_transformer_blocks, P.frame_seq_length = 2, 4
    P.randn_like = TD.HashRandn(5)
    if rec["switches"] is None:
        _, lat = P.inference(noise, ["p0"], return_latents=True)
    else:
        prompts = [[f"p{i}"] for i in range(len(rec["switches"]) + 1)]
        _, lat = P.inference(noise, text_prompts_list=prompts, switch_frame_indices=rec["switches"], return_latents=True)
    assert len(fg.log) == len(rec["log"])
    for i, (a, b) in enumerate(zip(fg.log, rec["log"])):
        assert a == b, f"call {i}: {a} != {b}"
    assert torch.equal(lat, rec["latents"])


@pytest.mark.parametrize("tag", ["plain", "last_step_ctx", "switch_mid", "switch_ext", "switch_none", "switch_at0"])
def test_training_rollout_call_sequence_matches_reference(tag):
    """Forward-only mirrors of the reference's training roll-out pipelines (requires_grad=False): same generator calls, outputs,
    return values and clear_kv_cache effects as pipeline/streaming_training.py / streaming_switch_training.py
    (tests/golden/train_calls.pt, oracle/make_golden.py::gen_train_calls)."""
    from oracle.make_golden import TRAIN_CASES, run_train_case
    from longlive_amd.pipeline import StreamingSwitchTrainingPipeline, StreamingTrainingPipeline
    rec = load_golden("train_calls.pt")[tag]
    _, kind, ctor, chunks = next(c for c in TRAIN_CASES if c[0] == tag)
    fg = FakeGenerator(R.FlowMatchSchedulerRef(5.0), 4)

    def patch(P):
        P.randn_like = TD.HashRandn(5)
    got = run_train_case({"train": StreamingTrainingPipeline, "switch": StreamingSwitchTrainingPipeline}, fg, kind, ctor, chunks,
                         synth.WanConfig(lat_h=4, lat_w=4), patch)
    assert len(got["log"]) == len(rec["log"])
    for i, (a, b) in enumerate(zip(got["log"], rec["log"])):
        assert a == b, f"call {i}: {a} != {b}"
    for a, b in zip(got["outs"], rec["outs"]):
        assert torch.equal(a, b)
    assert got["infos"] == rec["infos"] and got["cleared"] == rec["cleared"]


def test_training_rollout_is_forward_only_and_reports_exit_steps():
    from longlive_amd.pipeline import StreamingTrainingPipeline
    sch = R.FlowMatchSchedulerRef(5.0)
    fg = FakeGenerator(sch, 4)
    steps = torch.tensor([1000, 750, 500, 250])
    P = StreamingTrainingPipeline(denoising_step_list=steps, scheduler=sch, generator=fg, same_step_across_blocks=True,
                                  last_step_only=True, local_attn_size=12)
    P.num_transformer_blocks, P.frame_seq_length, P.kv_cache_size = 2, 4, 33 * 4
    P._initialize_kv_cache(1, torch.bfloat16, "cpu")
    P._initialize_crossattn_cache(1, torch.bfloat16, "cpu")
    noise = synth.synth_noise(synth.WanConfig(lat_h=4, lat_w=4), 3, seed=3)
    cond = {"prompt_embeds": torch.zeros(1, 1), "name": "p"}
    with pytest.raises(NotImplementedError, match="forward-only"):
        P.generate_chunk_with_cache(noise, cond)                      # requires_grad defaults to True upstream
    out, t_from, t_to, sim = P.generate_chunk_with_cache(noise, cond, requires_grad=False, return_sim_step=True)
    want_from = 1000 - int(torch.argmin((sch.timesteps.float() - 250.0).abs()))     # streaming_training.py:231-235
    assert (t_from, t_to, sim) == (want_from, 0, 4) and len(fg.log) == 5
    assert P.generator.model.max_attention_size == 12 * 4


def test_stream_yields_blocks_and_matches_inference():
    cfg = synth.WanConfig(lat_h=4, lat_w=4)
    enc = lambda text_prompts: {"prompt_embeds": torch.zeros(1, 1), "name": text_prompts[0]}
    noise = synth.synth_noise(cfg, 12, seed=3)
    outs = []
    for mode in ("inference", "stream"):
        fg = FakeGenerator(R.FlowMatchSchedulerRef(5.0), 4)
        P = CausalInferencePipeline(_args(True), "cpu", generator=fg, text_encoder=enc)
        P.num_transformer_blocks, P.frame_seq_length = 2, 4
        P.randn_like = TD.HashRandn(5)
        if mode == "inference":
            _, lat = P.inference(noise, ["p0"], return_latents=True)
        else:
            lat = torch.zeros_like(noise)
            starts = [s for s, _ in P.stream(noise, ["p0"], output=lat)]
            assert starts == [0, 3, 6, 9]
        outs.append(lat)
    assert torch.equal(outs[0], outs[1])


def test_errors():
    fg = FakeGenerator(R.FlowMatchSchedulerRef(5.0), 4)
    P = CausalInferencePipeline(_args(True), "cpu", generator=fg)
    noise = synth.synth_noise(synth.WanConfig(lat_h=4, lat_w=4), 4, seed=3)
    with pytest.raises(AssertionError):          # 4 % 3 != 0  (causal_inference.py:77)
        P.inference(noise, {"prompt_embeds": torch.zeros(1, 1), "name": "p"})
    with pytest.raises(RuntimeError, match="text_encoder"):
        P.inference(noise[:, :3], ["a prompt"])
    I = InteractiveCausalInferencePipeline(_args(False), "cpu", generator=fg)
    with pytest.raises(AssertionError):          # interactive_causal_inference.py:131-133
        I.inference(noise[:, :3], text_prompts_list=[["a"], ["b"]], switch_frame_indices=[])
