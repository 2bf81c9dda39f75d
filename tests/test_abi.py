"""CPU-side checks of the C-ABI boundary: the shared library loads without a GPU, exports every symbol that
include/longlive_hip.h declares, the ctypes table matches the header's parameter counts, and argument validation
fails loudly (status code + message) BEFORE anything is launched."""
import ctypes as C
import os
import re

import pytest

from longlive_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "longlive_hip.h")).read()


def _declarations():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|long long|const char\*)\s+(ll_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", body, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return decls


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    decls = _declarations()
    assert len(decls) >= 15
    for name in decls:
        assert hasattr(lib, name), f"{name} declared in include/longlive_hip.h but not exported"
    want = int(re.search(r"#define\s+LL_ABI_VERSION\s+(\d+)", HEADER).group(1))
    assert lib.ll_version() == want == _lib.ABI_VERSION      # a stale .so or binding is refused at load (_lib.load)


def test_ctypes_table_matches_header():
    decls = _declarations()
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, n in decls.items():
        assert len(_lib.SIGNATURES[name]) == n, f"{name}: header has {n} parameters, ctypes table {len(_lib.SIGNATURES[name])}"


def test_no_torch_types_in_the_abi():
    code = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)          # comments may mention torch; declarations may not
    assert "torch" not in code and "at::" not in code and "#include <hip" not in code
    assert re.findall(r"#include\s*<([^>]+)>", code) == ["stdint.h"]


@pytest.mark.parametrize("call,needle", [
    (lambda L: L.ll_gemm_bf16(0, 0, 1, 0, 128, 128, 100, 100, 128, 0, 0, 0, 0, 0, 0, 0, 0, None), "K=100"),
    (lambda L: L.ll_gemm_bf16(0, 0, 0, 0, 128, 128, 128, 128, 128, 0, 0, 0, 0, 0, 0, 0, 0, None), "bias"),
    (lambda L: L.ll_gemm_bf16(0, 0, 1, 0, 128, 128, 128, 128, 128, 7, 0, 0, 0, 0, 0, 0, 0, None), "epilogue"),
    (lambda L: L.ll_ln_modulate(0, 0, 0, 0, 6, 0, 1, 1, 10, 1536, 3, 1e-6, None), "not divisible"),
    (lambda L: L.ll_ln_modulate(0, 0, 0, 0, 6, 0, 1, 1, 9, 4096, 3, 1e-6, None), "C=4096"),
    (lambda L: L.ll_kv_roll(0, 0, 1, 100, 1536, 10, 5, 20, None), "src > dst"),
    (lambda L: L.ll_kv_roll(0, 0, 1, 100, 1536, 5, 90, 20, None), "outside cache"),
    (lambda L: L.ll_qk_norm_rope_kv_store(0, 0, 0, 0, 0, 0, 0, 0, 1, 4680, 1536, 128, 1560, 1023, 18720, 0, 0, 4680, 1e-6, None), "RoPE table"),
    (lambda L: L.ll_qk_norm_rope_kv_store(0, 0, 0, 0, 0, 0, 0, 0, 1, 4680, 1536, 128, 1560, 0, 18720, 18000, 0, 4680, 1e-6, None), "outside cache"),
    (lambda L: L.ll_flash_attn(0, 0, 0, 0, 1, 128, 12, 1536, 1536, 1536, 0, 0, 0, 0, 0, 0.088, None), "non-empty"),
    (lambda L: L.ll_flash_attn_qnorm(1, 1, 1, 1e-6, 1, 1, 1, 1, 128, 12, 1536, 1536, 1536, 0, 0, 16, 0.088, None), "not covered"),
    (lambda L: L.ll_flash_attn_qnorm(1, 1, 1, 1e-6, 1, 1, 1, 1, 128, 12, 3072, 1536, 1536, 0, 0, 512, 0.088, None), "whole projection output"),
    (lambda L: L.ll_gemm_bf16_ssq(1, 1, 1, 1, 4, 300, 136, 256, 256, 136, None), "not covered"),
    (lambda L: L.ll_synth_hash(None, 0, 10, 1, 2, None), "kind"),
    (lambda L: L.ll_linear_small(0, 0, 0, 0, 9, 64, 64, 0, 0, None), "M=9"),
    (lambda L: L.ll_gemm_bf16_qkv(0, 0, 1, 0, 128, 384, 128, 128, 384, 0, 1, 128, 64, 0, 0, 128, None), "cache_v"),
    (lambda L: L.ll_gemm_bf16_qkv(0, 0, 1, 0, 128, 384, 128, 128, 384, 1, 1, 100, 64, 0, 0, 10, None), "not B"),
    (lambda L: L.ll_gemm_bf16_qkv(0, 0, 1, 0, 128, 384, 128, 128, 384, 1, 1, 128, 64, 60, 0, 10, None), "outside cache"),
    (lambda L: L.ll_gemm_bf16(0, 0, 1, 0, 128, 128, 128, 128, 128, 2, 0, 0, 0, 6, 2, 128, 64, None), "res and e"),
    (lambda L: L.ll_modulation_table(0, 0, 0, 30, 3, 6, 1537, None), "bad shape"),
    (lambda L: L.ll_modulation_table_f32(0, 0, 0, 30, 3, 6, 1536, 0b1000000, None), "one_plus_mask"),
    (lambda L: L.ll_ln_modulate_tab(0, 1, 1, 1, 0, 6, 0, 1, 1, 9, 1536, 3, 1e-6, None), "exactly one"),
    (lambda L: L.ll_ln_modulate_tab(0, 1, 0, 0, 0, 6, 0, 7, 1, 9, 1536, 3, 1e-6, None), "bad mod index"),
    (lambda L: L.ll_conv_cl(0, 1, 1, 1, 0, 1, 2, 16, 32, 96, 96, 2624, 3, 3, 0, 96, None), "null operand"),
    (lambda L: L.ll_gemm_bf16_ksplit(0, 0, 1, 0, 512, 4096, 4096, 4096, 4096, 1, 0, None, 0, None), "bias or bias + residual"),
    (lambda L: L.ll_gemm_bf16_ksplit(0, 0, 1, 0, 512, 4096, 4096, 4096, 4096, 3, 0, None, 0, None), "needs res"),
    (lambda L: L.ll_gemm_bf16_ksplit(0, 0, 1, 0, 512, 4096, 4096, 4096, 4096, 0, 0, None, 64, None), "without a workspace"),
    (lambda L: L.ll_gemm_bf16_ksplit_t5norm(0, 0, 1, 0, 512, 4096, 4096, 4096, 4096, 1, None, 1e-6, None, None, 0, None), "norm weight"),
    (lambda L: L.ll_gemm_bf16_ksplit_t5norm(0, 0, 1, 0, 512, 4096, 4096, 4096, 4100, 1, 1, 1e-6, 1, None, 0, None), "must equal N"),
])
def test_invalid_arguments_are_rejected_before_launch(call, needle):
    lib = _lib.load()
    rc = call(lib)
    assert rc == -1, rc                       # LL_ERR_INVALID_ARG
    msg = lib.ll_last_error().decode()
    assert needle in msg, msg
    with pytest.raises(RuntimeError):
        _lib.check(rc, "test")


def test_small_m_split_k_plan_without_a_gpu():
    """Host-only helper of the small-M split-K path: not planned without a device (the CU count decides)."""
    lib = _lib.load()
    import torch
    if not torch.cuda.is_available():
        assert lib.ll_gemm_ksplit_plan(512, 4096, 4096) == 0 and lib.ll_gemm_ksplit_workspace_bytes(512, 4096, 4096) == 0


def test_gemm_plan_names_the_kernel_family_a_call_takes():
    """ll_gemm_plan_epi (host only): under the shipped tuning the six block linears of the pipeline take the generated kernels
    (gemm_asm_<width>_<epilogue>), int8 / modulation-vector calls and unsupported widths the HIP ones; tuning key gemm_asm = 0
    restores the HIP set; unknown tuning keys are rejected."""
    import ctypes as C
    lib = _lib.load()
    buf = C.create_string_buffer(256)

    def plan(M, N, K, i8, epi, plain):
        _lib.check(lib.ll_gemm_plan_epi(M, N, K, i8, epi, plain, buf, 256), "plan")
        return buf.value.decode()

    L, Cw, F1 = 4680, 1536, 8960
    assert plan(L, F1, Cw, 0, 1, 1).startswith("gemm_asm_224_gelu")
    assert plan(L, 3 * Cw, Cw, 0, 0, 2).startswith("gemm_asm_192_bias")            # fused QKV, one batch element
    assert plan(L, Cw, Cw, 0, 2, 1).startswith("gemm_asm_128_gate_res")
    assert plan(L, Cw, Cw, 0, 3, 1).startswith("gemm_asm_128_res")
    assert plan(L, Cw, Cw, 0, 0, 1).startswith("gemm_asm_128_bias")
    assert "760 workgroups" in plan(L, F1, Cw, 0, 1, 1) and "456 workgroups" in plan(L, 3 * Cw, Cw, 0, 0, 2)
    for text in (plan(L, Cw, Cw, 1, 2, 1), plan(L, Cw, Cw, 0, 2, 0), plan(L, 1000, Cw, 0, 0, 1), plan(L, F1, 192, 0, 1, 1)):
        assert text.startswith("gemm_kernel_v"), text                                # int8; modulation vector; N % 128; K < 256
    assert lib.ll_gemm_plan_epi(L, Cw, Cw, 0, 0, 1, None, 0) == -1
    try:
        assert lib.ll_set_tuning(b"gemm_asm", 0) == 0
        assert plan(L, F1, Cw, 0, 1, 1).startswith("gemm_kernel_v5")
    finally:
        assert lib.ll_set_tuning(b"gemm_asm", 35) == 0
    for key in (b"attn_asm", b"attn_asm_min_keys", b"gemm_asm"):
        assert lib.ll_set_tuning(key, {b"attn_asm": 1, b"attn_asm_min_keys": 512, b"gemm_asm": 35}[key]) == 0, key
    for gone in (b"no_such_key", b"attn_mfma16", b"attn_sk_wgs", b"gemm_ws", b"gemm_splitk_l2"):      # pruned in round 4: experiments/
        assert lib.ll_set_tuning(gone, 1) == -1, gone


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: handing the product path a host tensor is an error, not a silent slow path."""
    import torch
    from longlive_amd import ops
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.gemm(x, x, x[0])
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layernorm_affine(x, x[0], x[0], 1e-6)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
