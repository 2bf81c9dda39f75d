"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (longlive_amd/).

Imports the *reference* modules from /root/reference on CPU so that golden
vectors can be generated in the build container.  /root/reference does not
exist on the GPU box: only oracle/make_golden.py (run by hand, in the build
container) uses this file.

The reference cannot be imported unmodified without network access; four local
shims are needed:
  1. stub `diffusers` base classes (ConfigMixin / register_to_config / ModelMixin)
     -- used only as base classes/decorator (wan/modules/causal_model.py:14,16,511,522).
  2. pre-register empty `wan` / `wan.modules` packages with the real __path__ so
     wan/__init__.py:1-3 and wan/modules/__init__.py:1-5 (easydict, ftfy,
     torchvision) do not run.
  3. torch.cuda.current_device -> 0 while utils/memory.py:9 is imported.
  4. wan.modules.model.flash_attention := wan.modules.attention.attention, because
     cross-attention calls flash_attention directly (wan/modules/model.py:189),
     which asserts CUDA + flash-attn (wan/modules/attention.py:73,130), while
     attention() has the SDPA fallback (wan/modules/attention.py:182-197).
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("LONGLIVE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True   # never write __pycache__ into the read-only reference tree


def _install_shims():
    # (1) diffusers stubs
    if "diffusers" not in sys.modules:
        d = types.ModuleType("diffusers")
        cu = types.ModuleType("diffusers.configuration_utils")
        mm = types.ModuleType("diffusers.models")
        mu = types.ModuleType("diffusers.models.modeling_utils")

        class ConfigMixin:
            pass

        def register_to_config(fn):
            return fn

        class ModelMixin(nn.Module):
            pass

        cu.ConfigMixin = ConfigMixin
        cu.register_to_config = register_to_config
        mu.ModelMixin = ModelMixin
        d.configuration_utils = cu
        d.models = mm
        mm.modeling_utils = mu
        sys.modules.update({
            "diffusers": d,
            "diffusers.configuration_utils": cu,
            "diffusers.models": mm,
            "diffusers.models.modeling_utils": mu,
        })
    # (2) bare `wan`, `wan.modules` packages
    #     (+ `pipeline`, whose __init__ imports the training pipelines)
    for name, sub in (("wan", "wan"), ("wan.modules", "wan/modules"), ("pipeline", "pipeline")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF_ROOT, sub)]
            sys.modules[name] = m
    if "ftfy" not in sys.modules:
        sys.modules["ftfy"] = types.ModuleType("ftfy")


def load_reference():
    """Returns a namespace with the reference's hot-path modules."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    _install_shims()
    # (3) utils/memory.py calls torch.cuda.current_device() at import
    orig = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0
    try:
        importlib.import_module("utils.memory")
    finally:
        torch.cuda.current_device = orig
    attention = importlib.import_module("wan.modules.attention")
    model = importlib.import_module("wan.modules.model")
    # (4) SDPA fallback for cross-attention
    model.flash_attention = attention.attention
    causal_model = importlib.import_module("wan.modules.causal_model")
    scheduler = importlib.import_module("utils.scheduler")
    ns = types.SimpleNamespace(
        attention=attention, model=model, causal_model=causal_model,
        scheduler=scheduler)
    return ns


def load_pipelines():
    """Reference pipeline + wrapper classes (pipeline/causal_inference.py,
    pipeline/interactive_causal_inference.py, utils/wan_wrapper.py)."""
    ns = load_reference()
    # shim (3) again: wan/modules/t5.py:478 evaluates torch.cuda.current_device() in a default argument
    orig = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0
    try:
        ns.wan_wrapper = importlib.import_module("utils.wan_wrapper")
        ns.causal_inference = importlib.import_module("pipeline.causal_inference")
        ns.interactive = importlib.import_module("pipeline.interactive_causal_inference")
    finally:
        torch.cuda.current_device = orig
    return ns
