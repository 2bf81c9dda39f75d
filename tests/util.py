import torch

bf = torch.bfloat16


def bf16_ulp_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 units-in-the-last-place between two bf16 tensors (monotone integer mapping)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a.to(bf)) - key(b.to(bf))).abs()


def assert_bf16_close(got, want, max_ulp=1, min_exact=0.99, what="", atol=None):
    """Every element within `max_ulp` bf16 ulps of `want` (or within `atol`: results of cancelling sums sit near zero,
    where ulp distance is meaningless; default atol = 2^-8 of the tensor's RMS), and >= min_exact bit-identical."""
    got, want = got.detach().cpu(), want.detach().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert torch.isfinite(got.float()).all(), f"{what}: non-finite output"
    d = bf16_ulp_distance(got, want)
    exact = (d == 0).float().mean().item()
    if atol is None:
        atol = want.float().pow(2).mean().sqrt().item() * 2 ** -8
    bad = (d > max_ulp) & ((got.float() - want.float()).abs() > atol)
    nbad = int(bad.sum())
    if nbad:
        i = bad.flatten().nonzero()[0].item()
        raise AssertionError(f"{what}: {nbad} elements off by > {max_ulp} ulp and > {atol:.2e} (first: got "
                             f"{got.flatten()[i].item()} want {want.flatten()[i].item()}; exact fraction {exact:.4f})")
    assert exact >= min_exact, f"{what}: only {exact:.4f} of elements bit-exact (< {min_exact})"


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()
