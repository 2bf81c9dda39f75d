import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: real-shape CPU oracle checks (minutes); set LONGLIVE_SLOW=1")
    # synthetic weights / inputs on the CPU through a host build of csrc/synth_hash.h: the same integers as synth's int64 tensor form
    # (tests/test_synth_hash.py holds the two against each other), ~100x faster; silently absent without g++
    try:
        from oracle import fast_hash
        fast_hash.install()
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), map_location="cpu", weights_only=False)


@pytest.fixture(scope="session", autouse=True)
def _tuning_override():
    """LL_TUNING_TEST=key=value,... applies ll_set_tuning before the GPU tests: the parity suite then checks that kernel
    variant (used when A/B-ing variants on the GPU box)."""
    spec = os.environ.get("LL_TUNING_TEST", "")
    if spec:
        from longlive_amd import _lib
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            _lib.check(_lib.load().ll_set_tuning(k.encode(), int(v)), "ll_set_tuning")
    yield
