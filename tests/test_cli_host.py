"""Host-side parity of the CLI layer (longlive_amd/cli.py) with the reference's entry points: config keys, prompt file
formats (utils/dataset.py), rank partition (DistributedSampler(shuffle=False, drop_last=True)), output names, the
inference_iter rule, and the built-in video writer.  No GPU."""
import json
import os

import pytest
import torch

from longlive_amd import cli

REF_INFER_YAML = """
profile: true
denoising_step_list:
- 1000
- 750
- 500
- 250
warp_denoising_step: true
num_frame_per_block: 3
model_name: Wan2.1-T2V-1.3B
model_kwargs:
  local_attn_size: 12
  timestep_shift: 5.0
  sink_size: 3
data_path: prompts.txt
output_folder: videos/long
inference_iter: 1
num_output_frames: 120
use_ema: false
seed: 0
num_samples: 1
save_with_index: true
global_sink: true
context_noise: 0
switch_frame_indices: 40, 80, 120, 160, 200
adapter:
  type: "lora"
  rank: 256
  alpha: 256
"""


def test_config_matches_reference_yaml_keys(tmp_path):
    """Same keys and access patterns as configs/longlive_inference.yaml / longlive_interactive_inference.yaml."""
    p = tmp_path / "c.yaml"
    p.write_text(REF_INFER_YAML)
    c = cli.load_config(str(p))
    assert c.denoising_step_list == [1000, 750, 500, 250] and c.warp_denoising_step is True
    assert c.model_kwargs.local_attn_size == 12 and c.model_kwargs.timestep_shift == 5.0 and c.model_kwargs.sink_size == 3
    assert c.num_output_frames == 120 and c.global_sink is True and c.adapter.rank == 256
    assert getattr(c, "profile", False) is True and c.get("missing", 7) == 7 and "seed" in c
    assert cli.parse_switch_frame_indices(c.switch_frame_indices) == [40, 80, 120, 160, 200]
    assert cli.parse_switch_frame_indices(40) == [40] and cli.parse_switch_frame_indices("7,16,") == [7, 16]


def test_text_dataset_formats(tmp_path):
    p = tmp_path / "p.txt"
    p.write_text("a cat on a mat  \nsecond prompt\t\n\nlast")
    d = cli.TextDataset(str(p), str(p))
    assert len(d) == 4 and d[0] == {"prompts": "a cat on a mat", "idx": 0, "extended_prompts": "a cat on a mat"}
    assert d[2]["prompts"] == "" and d[3]["prompts"] == "last"           # the reference keeps empty lines (rstrip only)
    q = tmp_path / "q.txt"
    q.write_text("only one\n")
    with pytest.raises(AssertionError):
        cli.TextDataset(str(p), str(q))
    j = tmp_path / "m.jsonl"
    j.write_text(json.dumps({"prompts": ["a", "b", "c"]}) + "\n" + json.dumps({"prompts": ["d", "e", "f"]}) + "\n")
    m = cli.MultiTextDataset(str(j))
    assert len(m) == 2 and m[1] == {"idx": 1, "prompts_list": ["d", "e", "f"]}
    j.write_text(json.dumps({"prompts": ["a", "b"]}) + "\n" + json.dumps({"prompts": ["d"]}) + "\n")
    with pytest.raises(AssertionError):
        cli.MultiTextDataset(str(j))
    j.write_text(json.dumps({"text": ["a"]}) + "\n")
    with pytest.raises(AssertionError):
        cli.MultiTextDataset(str(j))
    j.write_text("")
    with pytest.raises(AssertionError):
        cli.MultiTextDataset(str(j))


@pytest.mark.parametrize("n,world", [(10, 4), (8, 8), (3, 2), (5, 1), (1, 2)])
def test_rank_partition_equals_distributed_sampler(n, world):
    from torch.utils.data.distributed import DistributedSampler
    for rank in range(world):
        if world == 1:
            want = list(range(n))
        else:
            want = list(DistributedSampler(list(range(n)), num_replicas=world, rank=rank, shuffle=False, drop_last=True))
        assert cli.rank_indices(n, rank, world) == want


def test_output_names_and_model_type():
    c = cli.Config(use_ema=False)
    assert cli.model_type_of(c, True) == "lora" and cli.model_type_of(c, False) == "regular"
    assert cli.model_type_of(cli.Config(use_ema=True), False) == "ema"
    assert cli.output_name(3, 17, 0, "lora", True, "x") == "rank3-17-0_lora.mp4"
    long = "p" * 150
    assert cli.output_name(0, 1, 2, "regular", False, long) == f"rank0-{'p' * 100}-2.mp4"
    assert cli.output_name(0, 1, 2, "ema", False, "a/b", interactive=True) == "rank0-a_b-2_ema.mp4"


def test_avi_writer_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    frames = torch.randint(0, 256, (5, 6, 7, 3), dtype=torch.uint8, generator=g)        # W*3 = 21: exercises row padding
    p = str(tmp_path / "v.avi")
    cli.write_avi_rgb24(p, frames, fps=16)
    assert torch.equal(cli.read_avi_rgb24(p), frames)
    raw = open(p, "rb").read()
    assert raw[:4] == b"RIFF" and int.from_bytes(raw[4:8], "little") == len(raw) - 8
    out = cli.write_video(str(tmp_path / "w.mp4"), frames.float(), fps=16)               # falls back to .avi without torchvision
    assert os.path.exists(out) and torch.equal(cli.read_avi_rgb24(out), frames) if out.endswith(".avi") else True


def test_hash_tokenizer_shapes():
    tok = cli.HashTokenizer(4096, 64)
    ids, mask = tok(["a cat  on a mat", "x"], return_mask=True, add_special_tokens=True)
    assert ids.shape == mask.shape == (2, 64) and ids.dtype == torch.long
    assert mask[0].sum() == 6 and ids[0, 5] == 1 and ids[0, 6] == 0 and mask[1].sum() == 2
    assert int(ids.max()) < 4096 and ids[0, 0] == ids[0, 3]                              # "a" twice -> same id
