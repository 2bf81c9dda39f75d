"""End-to-end CLI (longlive_amd/cli.py) on the GPU in synthetic mode, shrunk: prompt file -> hash tokenizer -> umT5 (1
layer) -> DiT (2 layers, 8x12 latents) -> VAE decoder -> video file, for both entry points."""
import json

import pytest
import torch

from longlive_amd import cli

pytestmark = pytest.mark.gpu


def _config(tmp_path, **kw):
    base = dict(profile=False, denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                model_kwargs=cli.Config(local_attn_size=12, timestep_shift=5.0, sink_size=3), output_folder=str(tmp_path / "out"),
                inference_iter=-1, num_output_frames=6, use_ema=False, seed=0, num_samples=1, save_with_index=True,
                global_sink=True, context_noise=0, synthetic=True,
                synthetic_overrides=cli.Config(num_layers=2, t5_layers=1, lat_h=8, lat_w=12))
    base.update(kw)
    return cli.Config(**base)


def test_cli_inference_synthetic(tmp_path):
    p = tmp_path / "prompts.txt"
    p.write_text("a cat on a mat\na red fox in the snow\n")
    recs = cli.run("inference", _config(tmp_path, data_path=str(p)), device=torch.device("cuda", 0))
    assert [r["idx"] for r in recs] == [0, 1] and all(r["frames"] == 21 for r in recs)
    assert recs[0]["path"].endswith(("rank0-0-0_regular.mp4", "rank0-0-0_regular.avi"))
    if recs[0]["path"].endswith(".avi"):
        v0, v1 = cli.read_avi_rgb24(recs[0]["path"]), cli.read_avi_rgb24(recs[1]["path"])
        assert v0.shape == (21, 64, 96, 3) and v0.float().std() > 1.0
        assert not torch.equal(v0, v1)                                   # different prompt + different noise


def test_cli_interactive_synthetic_and_inference_iter(tmp_path):
    j = tmp_path / "m.jsonl"
    j.write_text("\n".join(json.dumps({"prompts": [f"scene {k} part {s}" for s in range(2)]}) for k in range(3)) + "\n")
    cfg = _config(tmp_path, data_path=str(j), switch_frame_indices="3", global_sink=False, inference_iter=0, num_output_frames=9)
    recs = cli.run("interactive", cfg, device=torch.device("cuda", 0))
    assert len(recs) == 1 and recs[0]["idx"] == 0 and recs[0]["frames"] == 33       # inference_iter = 0: stop after the first line
    cfg.switch_frame_indices = "3, 6"
    with pytest.raises(AssertionError):
        cli.run("interactive", cfg, device=torch.device("cuda", 0))                 # 2 segments need exactly 1 switch index
