"""Parity of the VAE-decoder kernels and of the assembled streaming decoder (longlive_amd/vae.py) against the CPU oracle
(oracle/ref_vae.py, pinned bit-exact to the reference's WanVAE_) and against the goldens the reference itself produced
(tests/golden/vae_decode.pt).  Everything goes through the C ABI."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from longlive_amd import synth
from util import assert_bf16_close, bf, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from longlive_amd import ops as o
    return o


def hn(name, shape, scale=1.0, shift=0.0, seed=77):
    return (synth.hash_normal(seed, name, shape) * scale + shift).to(bf)


def to_cl(x):       # [1, C, T, H, W] -> [T, H, W, C]
    return x[0].permute(1, 2, 3, 0).contiguous()


@pytest.mark.parametrize("T,H,W,Cin,Cout,KT,KH,up,with_res", [
    (1, 8, 12, 16, 384, 3, 3, False, False),     # decoder.conv1: K = 432 -> padded to 448, NT = 4
    (2, 9, 7, 96, 96, 3, 3, False, True),        # NT = 3 tile, ragged M, fused residual
    (1, 6, 10, 192, 192, 3, 3, False, True),
    (3, 5, 6, 384, 768, 3, 1, False, False),     # time_conv (3,1,1)
    (2, 6, 5, 384, 192, 1, 3, True, False),      # Upsample(nearest x2) + Conv2d 3x3
    (2, 7, 9, 192, 96, 1, 1, False, False),      # shortcut 1x1x1
    (1, 16, 24, 96, 3, 3, 3, False, False),      # head: Cout 3 -> padded to 8, NT = 1 tile
    (1, 4, 4, 16, 16, 1, 1, False, False),       # conv2: K = 16 -> one padded k-step
    (2, 40, 50, 96, 96, 3, 3, False, False),     # several tiles per XCD
    (2, 32, 64, 96, 96, 3, 3, False, True),      # H % 16 == 0, W % 32 == 0: the halo-tile kernel, 2 x 2 tiles per frame, residual
    (1, 16, 32, 192, 192, 3, 3, False, False),   # halo kernel: one tile (all four borders), six channel slices, two N tiles
    (3, 48, 96, 96, 96, 3, 3, False, False),     # halo kernel: interior tiles, three frames
    (2, 32, 32, 96, 3, 3, 3, False, False),      # halo kernel, head: one 16-channel block fed by the 8 padded weight rows
    (2, 8, 16, 384, 192, 1, 3, True, False),     # halo kernel, upsampled 1x3x3: one output tile (16 x 32), twelve slices
    (1, 24, 48, 192, 96, 1, 3, True, False),     # halo kernel, upsampled: 3 x 3 output tiles
    (2, 30, 52, 384, 384, 3, 3, False, False),   # halo kernel, partial tiles at the right and bottom edge (60 x 104 in small), 4 N tiles
    (1, 30, 52, 96, 96, 3, 3, False, True),      # ... with the fused residual
    (1, 15, 26, 192, 96, 1, 3, True, False),     # halo kernel, upsampled, partial edge tiles (output 30 x 52)
])
def test_conv_cl(ops, T, H, W, Cin, Cout, KT, KH, up, with_res):
    x = hn("cx", (1, Cin, T, H, W))
    cache = hn("cc", (1, Cin, 2, H, W)) if KT == 3 else None
    w = hn("cw", (Cout, Cin, KT, KH, KH), 1.0 / math.sqrt(Cin * KT * KH * KH))
    b = hn("cb", (Cout,), 0.1)
    xin = torch.cat([cache, x], 2) if KT == 3 else x
    xf = xin.float()
    if up:
        xf = F.interpolate(xf[0].permute(1, 0, 2, 3), scale_factor=(2.0, 2.0), mode="nearest").permute(1, 0, 2, 3)[None]
    p = KH // 2
    want = F.conv3d(F.pad(xf, (p, p, p, p, 0, 0)), w.float(), b.float()).to(bf)        # [1, Cout, T, Ho, Wo]
    res = hn("cr", tuple(want.shape)) if with_res else None
    if with_res:
        want = (want.float() + res.float()).to(bf)
    pk, pb, geo = ops.pack_conv_weight(w.to(DEV), b.to(DEV))
    res_cl = None
    if with_res:
        res_cl = torch.zeros(T, want.shape[3], want.shape[4], geo[1], dtype=bf, device=DEV)
        res_cl[..., :Cout] = to_cl(res).to(DEV)
    got = ops.conv_cl(to_cl(xin).to(DEV), pk, pb, geo, upsample=up, res=res_cl)      # history frames first when KT == 3
    torch.cuda.synchronize()
    assert got.shape[-1] == geo[1]
    # fp32 accumulation in a different order than the CPU conv: <= 1 ulp apart nearly everywhere
    assert_bf16_close(got[..., :Cout], to_cl(want), 2, 0.97, f"conv_cl {Cin}->{Cout} k{KT}x{KH}x{KH} up={up}")


def test_conv_halo_and_implicit_gemm_kernels_agree(ops):
    """Same convolution through both kernels (tuning key conv_halo): fp32 sums in a different K order, <= 1 bf16 ulp apart."""
    from longlive_amd import _lib
    T, H, W, C = 2, 32, 64, 96
    x = hn("hx", (1, C, T + 2, H, W))
    w = hn("hw", (C, C, 3, 3, 3), 1.0 / math.sqrt(C * 27))
    pk, pb, geo = ops.pack_conv_weight(w.to(DEV), hn("hb", (C,), 0.1).to(DEV))
    outs = []
    try:
        for v in (1, 0):
            assert _lib.load().ll_set_tuning(b"conv_halo", v) == 0
            outs.append(ops.conv_cl(to_cl(x).to(DEV), pk, pb, geo))
    finally:
        _lib.load().ll_set_tuning(b"conv_halo", 1)
    torch.cuda.synchronize()
    assert_bf16_close(outs[0], outs[1], 1, 0.9, "halo vs implicit GEMM")


@pytest.mark.parametrize("T,H,W,Cin,up,with_res,silu,want_raw", [
    (2, 32, 64, 96, False, False, True, False),      # ResidualBlock: conv1 -> RMS_norm -> SiLU (the un-normalised tensor is not kept)
    (1, 30, 52, 96, False, True, True, True),        # conv2 + shortcut -> next block's RMS_norm, partial edge tiles, both outputs
    (2, 15, 26, 192, True, False, True, True),       # upsampled 1x3x3 (192 -> 96) feeding the first block of the 96-channel stage
    (1, 16, 32, 96, False, False, False, True),      # RMS_norm without SiLU
])
def test_conv_cl_rms_is_conv_then_rms_silu(ops, T, H, W, Cin, up, with_res, silu, want_raw):
    """ll_conv_cl_rms (the convolution's epilogue also applies the RMS_norm + SiLU that follows it, vae.py:193-220) against the two
    launches it replaces: the un-normalised output bit-identical (same accumulators, same epilogue arithmetic), the normalised one
    within 1 bf16 ulp (the sum of squares is taken in another order: 4 channels of each 16-block per lane instead of 8 consecutive)."""
    Cout, KT = 96, (1 if up else 3)
    x = hn("fx", (1, Cin, T + (2 if KT == 3 else 0), H, W))
    w = hn("fw", (Cout, Cin, KT, 3, 3), 1.0 / math.sqrt(Cin * KT * 9))
    g = hn("fg", (Cout,), 0.1, 1.0).to(DEV)
    pk, pb, geo = ops.pack_conv_weight(w.to(DEV), hn("fb", (Cout,), 0.1).to(DEV))
    assert ops.conv_cl_rms_ok(geo, H, W, up)
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    res = hn("fr", (T, Ho, Wo, Cout)).to(DEV) if with_res else None
    xin = to_cl(x).to(DEV)
    raw = ops.conv_cl(xin, pk, pb, geo, upsample=up, res=res)
    want = ops.rms_silu_cl(raw, g, silu=silu)
    out2 = torch.full((T, Ho, Wo, Cout), float("nan"), dtype=bf, device=DEV)
    got_raw = ops.conv_cl_rms(xin, pk, pb, geo, g, out2, silu=silu, upsample=up, res=res, want_raw=want_raw)
    torch.cuda.synchronize()
    if want_raw:
        assert torch.equal(got_raw, raw)
    else:
        assert got_raw is None
    assert_bf16_close(out2, want, 1, 0.98, f"conv + rms_silu fused, Cin={Cin} up={up} res={with_res}")
    # not covered: another channel count / the implicit-GEMM shapes -> the predicate says so and the entry point refuses
    pk2, pb2, geo2 = ops.pack_conv_weight(hn("fw2", (192, 192, 3, 3, 3)).to(DEV), torch.zeros(192, dtype=bf, device=DEV))
    assert not ops.conv_cl_rms_ok(geo2, 16, 32)
    with pytest.raises(RuntimeError):
        ops.conv_cl_rms(torch.zeros(3, 16, 32, 192, dtype=bf, device=DEV), pk2, pb2, geo2, torch.ones(192, dtype=bf, device=DEV),
                        torch.empty(1, 16, 32, 192, dtype=bf, device=DEV))


def test_conv_cl_rejects_bad_shapes(ops):
    w = hn("w", (96, 96, 3, 3, 3)).to(DEV)
    pk, pb, geo = ops.pack_conv_weight(w, torch.zeros(96, dtype=bf, device=DEV))
    with pytest.raises(AssertionError):
        ops.conv_cl(torch.zeros(2, 4, 4, 96, dtype=bf, device=DEV), pk, pb, geo)   # temporal conv without room for its history
    with pytest.raises(AssertionError):
        ops.conv_cl(torch.zeros(3, 4, 4, 64, dtype=bf, device=DEV), pk, pb, geo)   # channel mismatch


@pytest.mark.parametrize("C,pixels,silu", [(96, 1000, True), (192, 333, True), (384, 96, True), (384, 50, False)])
def test_rms_silu_cl(ops, C, pixels, silu):
    from oracle import ref_vae as RV
    x = hn("rx", (pixels, C), 1.3, 0.2)
    g = hn("rg", (C,), 0.1, 1.0)
    want = RV.rms_norm(x.t()[None, :, None, :, None].contiguous(), g.view(C, 1, 1, 1))    # channel-first [1,C,1,P,1]
    if silu:
        want = F.silu(want)
    want = want[0, :, 0, :, 0].t()
    got = ops.rms_silu_cl(x.to(DEV), g.to(DEV), silu=silu)
    assert_bf16_close(got, want, 1, 0.98, f"rms_silu C={C}")


def test_softmax_rows(ops):
    s = hn("sm", (70, 128), 8.0)
    got = ops.softmax_rows(s.to(DEV), 0.3, n_valid=96)
    want = torch.softmax(s[:, :96].float() * 0.3, -1)
    assert torch.all(got[:, 96:] == 0)
    assert rel_l2(got[:, :96].float().cpu(), want) < 4e-3
    assert torch.allclose(got.float().sum(-1).cpu(), torch.ones(70), atol=2e-2)


def test_unscale_and_clamp_layouts(ops):
    from oracle import ref_vae as RV
    z = hn("uz", (3, 16, 5, 7))
    mean = torch.tensor(RV.VAE_MEAN).to(bf)
    inv_std = 1.0 / torch.tensor(RV.VAE_STD).to(bf)
    want = (z / inv_std.view(1, -1, 1, 1) + mean.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    got = ops.vae_unscale_cl(z.to(DEV), mean.to(DEV), inv_std.to(DEV))
    assert torch.equal(got.cpu(), want.contiguous())
    y = hn("cy", (2, 6, 4, 8), 1.5)
    got = ops.cl_to_tchw_clamp(y.to(DEV))
    assert torch.equal(got.cpu(), y[..., :3].float().clamp(-1, 1).permute(0, 3, 1, 2))


@pytest.fixture(scope="module")
def vae():
    from longlive_amd.vae import WanVAEWrapper
    m = WanVAEWrapper(device=DEV, chunk=2)
    m.load_state_dict(synth.synth_vae_state_dict(synth.VaeConfig(), seed=5))
    return m


def test_vae_decode_matches_reference_golden(vae):
    """Full decode of 5 latent frames at 8x12 against the reference WanVAEWrapper.decode_to_pixel output.  ~45 bf16 layers
    deep; tolerance: rel-L2 <= 3e-2 on pixels in [-1, 1] (the oracle itself reproduces the golden bit-exactly on CPU)."""
    rec = load_golden("vae_decode.pt")
    lat = synth.hash_normal(55, "vae.latent", (1, 5, 16, 8, 12)).to(bf)
    got = vae.decode_to_pixel(lat.to(DEV), use_cache=False).cpu()
    want = rec["full"].float()
    assert got.shape == want.shape == (1, 17, 3, 64, 96)
    err = rel_l2(got, want)
    print(f"vae full decode rel-L2 {err:.3e}")
    assert err < 3e-2
    assert rel_l2(got[0, :, :, ::7, ::11], rec["full_f32_sample"]) < 3e-2


def test_vae_streaming_matches_reference_golden_and_chunking_is_exact(vae):
    rec = load_golden("vae_decode.pt")
    lat = synth.hash_normal(55, "vae.latent", (1, 5, 16, 8, 12)).to(bf).to(DEV)
    vae.model.clear_cache()
    a = vae.decode_to_pixel(lat[:, :2], use_cache=True)
    b = vae.decode_to_pixel(lat[:, 2:], use_cache=True)
    vae.model.clear_cache()
    assert a.shape == (1, 5, 3, 64, 96) and b.shape == (1, 12, 3, 64, 96)
    assert rel_l2(a.cpu(), rec["stream_a"].float()) < 3e-2
    assert rel_l2(b.cpu(), rec["stream_b"].float()) < 3e-2
    full = vae.decode_to_pixel(lat, use_cache=False)
    assert torch.equal(torch.cat([a, b], 1), full), "streamed pieces must reproduce the one-shot decode bit for bit"
    # frame-at-a-time (the reference's schedule) vs chunked launches: identical accumulation order per pixel
    vae.model.chunk = 1
    one = vae.decode_to_pixel(lat, use_cache=False)
    vae.model.chunk = 3
    three = vae.decode_to_pixel(lat, use_cache=False)
    vae.model.chunk = 2
    assert torch.equal(one, full) and torch.equal(three, full)


def test_vae_decode_matches_oracle_on_other_geometry(vae):
    """Seeded latents at a non-square, non-multiple-of-tile geometry against the CPU oracle (run here on the host)."""
    from oracle import ref_vae as RV
    vcfg = synth.VaeConfig()
    _, layers = synth.vae_decoder_layout(vcfg)
    dec = RV.RefVaeDecoder(synth.synth_vae_state_dict(vcfg, seed=5), layers)
    lat = synth.hash_normal(56, "vae.latent2", (1, 3, 16, 6, 10)).to(bf)
    want = RV.decode_to_pixel(dec, lat, use_cache=False)
    got = vae.decode_to_pixel(lat.to(DEV), use_cache=False).cpu()
    assert got.shape == want.shape
    assert rel_l2(got, want) < 3e-2


def test_pipeline_streams_pixels_identical_to_one_shot_decode(vae):
    """DiT (2 layers, 8x12 latents) + VAE through the pipeline: frames emitted live per block (streaming cache) are bit
    identical to `inference()`'s decode of the finished latents, and the latents equal the VAE-less run."""
    import test_model_gpu as TM
    from longlive_amd.pipeline import CausalInferencePipeline
    cfg, gen, enc = TM._pipe_generator()
    noise = synth.synth_noise(cfg, 9, seed=41, device=DEV)
    P = CausalInferencePipeline(TM._pipe_args(), DEV, generator=gen, text_encoder=enc, vae=vae)
    P.randn_like = TM.TD.HashRandn(43)
    video, lat = P.inference(noise, ["p0"], return_latents=True)
    assert video.shape == (1, 33, 3, 64, 96) and float(video.min()) >= 0.0 and float(video.max()) <= 1.0
    want = (vae.decode_to_pixel(lat, use_cache=False) * 0.5 + 0.5).clamp(0, 1)
    assert torch.equal(video, want)
    P.randn_like = TM.TD.HashRandn(43)
    pieces = [px for _, px in P.stream_video(noise, ["p0"])]
    assert [p.shape[1] for p in pieces] == [9, 12, 12]
    assert torch.equal(torch.cat(pieces, 1), video)
    # inference() with the per-block side-stream decode: the same video
    P.randn_like = TM.TD.HashRandn(43)
    P.overlap_decode = True
    video2, lat2 = P.inference(noise, ["p0"], return_latents=True)
    P.overlap_decode = False
    torch.cuda.synchronize()
    assert torch.equal(video2, video) and torch.equal(lat2, lat)
    # decode on a second stream beside the next block's generation: same frames, same order, one block later
    P.randn_like = TM.TD.HashRandn(43)
    got = [(st, px) for st, px in P.stream_video(noise, ["p0"], overlap_decode=True)]
    torch.cuda.synchronize()
    assert [st for st, _ in got] == [0, 3, 6]
    assert torch.equal(torch.cat([px for _, px in got], 1), video)


def test_vae_first_frame_at_real_resolution_matches_oracle(vae):
    """One latent frame at 60x104 -> 480x832 against the CPU oracle (a few seconds on the host): exercises the real tile
    counts (399360-pixel convolutions, 6240-token attention, every upsample) that the small goldens do not."""
    from oracle import ref_vae as RV
    vcfg = synth.VaeConfig()
    _, layers = synth.vae_decoder_layout(vcfg)
    dec = RV.RefVaeDecoder(synth.synth_vae_state_dict(vcfg, seed=5), layers)
    lat = synth.hash_normal(57, "vae.latent3", (1, 1, 16, 60, 104)).to(bf)
    want = RV.decode_to_pixel(dec, lat, use_cache=False)
    got = vae.decode_to_pixel(lat.to(DEV), use_cache=False).cpu()
    assert got.shape == want.shape == (1, 1, 3, 480, 832)
    r = rel_l2(got, want)
    print(f"vae 480x832 first frame rel-L2 {r:.2e}")
    assert r < 3e-2
