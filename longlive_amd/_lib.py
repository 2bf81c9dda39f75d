"""ctypes binding of liblonglive_hip.so (the C ABI declared in include/longlive_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol is absent, import of the ops
fails loudly (RuntimeError) rather than silently running something else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblonglive_hip.so")

ABI_VERSION = 106      # include/longlive_hip.h: LL_ABI_VERSION (tests/test_abi.py holds the two together)

_p, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong

# name -> argtypes (restype is always int unless listed in _RESTYPES); mirrors include/longlive_hip.h 1:1
SIGNATURES = {
    "ll_version": [],
    "ll_last_error": [],
    "ll_set_tuning": [C.c_char_p, _i],
    "ll_gemm_plan": [_i, _i, _i, _i, C.c_char_p, _i],
    "ll_gemm_plan_epi": [_i, _i, _i, _i, _i, _i, C.c_char_p, _i],
    "ll_flash_attn_plan": [_i, _i, _i, _i, _i, _i, C.c_char_p, _i],
    "ll_ln_modulate": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "ll_modulation_table": [_p, _p, _p, _i, _i, _i, _i, _p],
    "ll_modulation_table_f32": [_p, _p, _p, _i, _i, _i, _i, C.c_uint, _p],
    "ll_ln_modulate_tab": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "ll_layernorm_affine": [_p, _p, _p, _p, _i, _i, _f, _p],
    "ll_ln_modulate_q8": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "ll_layernorm_affine_q8": [_p, _p, _p, _p, _p, _i, _i, _f, _p],
    "ll_rmsnorm": [_p, _p, _p, _i, _i, _i, _i, _f, _p],
    "ll_qk_norm_rope_kv_store": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "ll_kv_roll": [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    "ll_gemm_bf16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p],
    "ll_gemm_ksplit_plan": [_i, _i, _i],
    "ll_gemm_ksplit_workspace_bytes": [_i, _i, _i],
    "ll_gemm_bf16_ksplit": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _ll, _p],
    "ll_gemm_bf16_ksplit_t5norm": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _f, _p, _p, _ll, _p],
    "ll_gemm_w8a8": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p],
    "ll_gemm_bf16_qkv": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "ll_gemm_w8a8_qkv": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p],
    "ll_quantize_rows": [_p, _p, _p, _i, _i, _i, _p],
    "ll_linear_small": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ll_flash_attn": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _ll, _i, _i, _i, _i, _f, _p],
    "ll_patchify": [_p, _p, _i, _i, _i, _i, _i, _p],
    "ll_sinusoid": [_p, _p, _i, _i, _p],
    "ll_unpatchify_x0": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ll_add_noise": [_p, _p, _p, _p, _i, _ll, _p],
    "ll_sigma_lookup": [_p, _p, _p, _p, _i, _i, _p],
    "ll_synth_hash": [_p, _ll, _ll, C.c_ulonglong, _i, _p],
    "ll_gemm_ssq_planes": [_i, _i, _i],
    "ll_gemm_bf16_ssq": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "ll_flash_attn_qnorm_ok": [_i, _i],
    "ll_flash_attn_qnorm": [_p, _p, _p, _f, _p, _p, _p, _i, _i, _i, _i, _i, _i, _ll, _i, _i, _f, _p],
    "ll_conv_cl": [_p] * 6 + [_i] * 10 + [_p],
    "ll_conv_cl_rms_ok": [_i] * 7,
    "ll_conv_cl_rms": [_p] * 8 + [_i] * 11 + [_p],
    "ll_rms_silu_cl": [_p, _p, _p, _ll, _i, _i, _p],
    "ll_softmax_rows": [_p, _p, _i, _i, _i, _f, _p],
    "ll_vae_unscale_cl": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ll_cl_to_tchw_clamp": [_p, _p, _i, _i, _i, _i, _p],
    "ll_t5_rmsnorm": [_p, _p, _p, _i, _i, _f, _p],
    "ll_t5_gated_gelu": [_p, _p, _ll, _i, _p],
    "ll_gather_rows": [_p, _p, _p, _i, _i, _ll, _p],
    "ll_t5_attention": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
}
_RESTYPES = {"ll_last_error": C.c_char_p, "ll_gemm_ksplit_workspace_bytes": C.c_longlong}

_lib = None


def load() -> C.CDLL:
    """Loads the HIP library (once).  Raises RuntimeError if it is missing: build it with
    `python -c 'import __graft_entry__ as g; g.build()'` or `make -C longlive_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("LONGLIVE_HIP_LIB", LIB_PATH)      # kernel A/B: another build of the same ABI
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built (make -C longlive_amd/csrc). "
            "longlive_amd has no CPU fallback by design.")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:  # pragma: no cover
            raise RuntimeError(f"{LIB_PATH} does not export {name}") from exc
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    if lib.ll_version() != ABI_VERSION:
        raise RuntimeError(f"{path} has ABI version {lib.ll_version()}, this binding needs {ABI_VERSION}: rebuild it "
                           "(make -C longlive_amd/csrc); a stale library would misread every argument list")
    _lib = lib
    for kv in filter(None, os.environ.get("LL_TUNING", "").split(",")):   # kernel A/B only, e.g. LL_TUNING=attn_variant=2,conv_halo=0
        k, v = kv.split("=")
        if lib.ll_set_tuning(k.encode(), int(v)) != 0:
            raise RuntimeError(f"LL_TUNING: {lib.ll_last_error().decode()}")
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ll_last_error()
        raise RuntimeError(f"longlive_hip {what} failed (status {rc}): {msg.decode() if msg else ''}")
