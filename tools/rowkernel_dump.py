#!/usr/bin/env python3
"""Outputs of the row kernels (ll_ln_modulate [+ q8], ll_layernorm_affine [+ q8], ll_rmsnorm, ll_qk_norm_rope_kv_store) on seeded
inputs at the production width and at a ragged one, written to a .pt file: run once per library build
(LONGLIVE_HIP_LIB=... python3 tools/rowkernel_dump.py out.pt) and compare the files (`--compare a.pt b.pt`: every tensor
bit-identical or the script exits 1).  Used when the arithmetic of these kernels is re-expressed without changing its values."""
import sys

import torch


def compare(a, b):
    A, B = torch.load(a), torch.load(b)
    bad = 0
    for k in sorted(A):
        same = torch.equal(A[k], B[k])
        n = A[k].numel()
        diff = 0 if same else int((A[k].view(torch.uint8 if A[k].dtype == torch.int8 else A[k].dtype) != B[k].view_as(A[k])).sum())
        print(f"{k:40s} {tuple(A[k].shape)!s:24s} {'identical' if same else f'{diff} of {n} differ'}")
        bad += not same
    sys.exit(1 if bad else 0)


def main(path):
    sys.path.insert(0, ".")
    from longlive_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(1234)
    out = {}

    def rnd(*shape, scale=1.0, dtype=torch.bfloat16):
        return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)

    for tag, (L, C, F, H) in {"prod": (4680, 1536, 3, 12), "ragged": (96, 1280, 3, 10), "small": (30, 256, 3, 2)}.items():
        x = rnd(1, L, C, scale=2.0) + 0.3
        e = rnd(1, F, 6, C, scale=0.5)
        mod = rnd(6, C, scale=0.5)
        out[f"{tag}.ln_mod_pre"] = ops.ln_modulate(x, e, None, 0, 1, F, 1e-6)
        out[f"{tag}.ln_mod"] = ops.ln_modulate(x, e, mod, 3, 4, F, 1e-6)
        qv, sc = ops.ln_modulate_q8(x, e, None, 0, 1, F, 1e-6)
        out[f"{tag}.ln_mod_q8"], out[f"{tag}.ln_mod_q8_scale"] = qv, sc
        w, b = rnd(C, scale=0.5) + 1.0, rnd(C, scale=0.2)
        out[f"{tag}.ln_affine"] = ops.layernorm_affine(x, w, b, 1e-6)
        qv, sc = ops.layernorm_affine_q8(x, w, b, 1e-6)
        out[f"{tag}.ln_affine_q8"], out[f"{tag}.ln_affine_q8_scale"] = qv, sc
        out[f"{tag}.rmsnorm"] = ops.rmsnorm(x, w, 1e-6)
        qkv = rnd(1, L, 3 * C, scale=1.5)
        fs = L // F
        nf = 64 - 2 * (64 // 3)
        ang = torch.rand(1024, nf, generator=g) * 6.28
        rf = torch.stack([ang.cos(), ang.sin()], -1).float().to(dev).contiguous()
        ang = torch.rand(fs, 64 - nf, generator=g) * 6.28
        rhw = torch.stack([ang.cos(), ang.sin()], -1).float().to(dev).contiguous()
        S = 2 * L
        ck = torch.zeros(1, S, H, 128, dtype=torch.bfloat16, device=dev)
        cv = torch.zeros_like(ck)
        q = torch.empty(1, L, C, dtype=torch.bfloat16, device=dev)
        ops.qk_norm_rope_kv_store(qkv, w, w.flip(0).contiguous(), rf, rhw, q, ck, cv, 128, fs, 5, L // 2, fs // 2, L - fs, 1e-6)
        out[f"{tag}.rope_q"], out[f"{tag}.rope_k"], out[f"{tag}.rope_v"] = q, ck, cv
    torch.cuda.synchronize()
    torch.save({k: v.cpu() for k, v in out.items()}, path)
    print(f"wrote {len(out)} tensors to {path}")


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        compare(sys.argv[2], sys.argv[3])
    main(sys.argv[1])
