"""The generators' timing-only knobs (ASM_NO_* / ASM_G_NO_*: a part of the loop removed, results INVALID) can never reach a shipped
kernel: the generators refuse them without --diag, and the Makefile runs the generators in an empty environment."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "longlive_amd", "csrc")
GEN = os.path.join(CSRC, "gen")


def _run(args, env_extra, cwd=CSRC):
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True)


def test_gemm_generator_refuses_timing_knobs_without_diag(tmp_path):
    out = tmp_path / "g.inc"
    r = _run([os.path.join(GEN, "gemm_asm_gen.py"), "128", "0", str(out)], {"ASM_G_NO_MFMA": "1"})
    assert r.returncode != 0 and "refused" in (r.stderr + r.stdout) and not out.exists()
    r = _run([os.path.join(GEN, "gemm_asm_gen.py"), "--diag", "128", "0", str(out)], {"ASM_G_NO_MFMA": "1"})
    assert r.returncode == 0 and out.exists()
    ref = tmp_path / "ref.inc"
    r = _run([os.path.join(GEN, "gemm_asm_gen.py"), "128", "0", str(ref)], {})
    assert r.returncode == 0
    assert out.read_text() != ref.read_text()                      # the knob does change the kernel: that is why it must not leak
    assert "v_mfma" not in out.read_text() and "v_mfma" in ref.read_text()


def test_attention_generator_refuses_timing_knobs_without_diag(tmp_path):
    out = tmp_path / "a.inc"
    r = _run([os.path.join(GEN, "attn_asm_gen.py"), str(out)], {"ASM_NO_EXP": "1"})
    assert r.returncode != 0 and "refused" in (r.stderr + r.stdout) and not out.exists()


def test_makefile_builds_the_same_kernels_under_a_stray_knob(tmp_path):
    """`make` under ASM_G_NO_MFMA=1 / ASM_NO_EXP=1 (as __graft_entry__.build() would pass them on from a polluted shell) generates
    byte-identical kernel text: the generator commands start from `env -i`."""
    r = subprocess.run(["make", "-C", CSRC, "-n", "-B", "build/gemm_asm_128_0.inc", "build/attn_asm_body.inc"], capture_output=True,
                       text=True, env=dict(os.environ, ASM_G_NO_MFMA="1", ASM_NO_EXP="1"))
    assert r.returncode == 0, r.stderr
    cmds = [l for l in r.stdout.splitlines() if "_asm_gen.py" in l]
    assert len(cmds) == 2 and all(l.startswith("env -i ") for l in cmds), cmds
    # run exactly those commands, redirected into tmp_path, with the stray variables set
    for l in cmds:
        l2 = l.replace("build/", str(tmp_path) + "/")
        rr = subprocess.run(l2, shell=True, cwd=CSRC, capture_output=True, text=True, env=dict(os.environ, ASM_G_NO_MFMA="1", ASM_NO_EXP="1"))
        assert rr.returncode == 0, rr.stderr
    for name, gen, args in (("gemm_asm_128_0.inc", "gemm_asm_gen.py", ["128", "0"]), ("attn_asm_body.inc", "attn_asm_gen.py", [])):
        clean = tmp_path / ("clean_" + name)
        env = {k: v for k, v in os.environ.items() if not k.startswith("ASM_")}
        rr = subprocess.run([sys.executable, os.path.join(GEN, gen)] + args + [str(clean)], cwd=CSRC, capture_output=True, text=True, env=env)
        assert rr.returncode == 0, rr.stderr
        assert (tmp_path / name).read_text() == clean.read_text(), name


def test_knobs_header_records_schedule_knobs(tmp_path):
    h = tmp_path / "k.h"
    r = _run([os.path.join(GEN, "attn_asm_gen.py"), "--knobs-header", str(h)], {"ASM_VF_LEAD": "2.5"})
    assert r.returncode == 0 and 'ASM_VF_LEAD=2.5#' in h.read_text()
    env = {k: v for k, v in os.environ.items() if not k.startswith("ASM_")}
    r = subprocess.run([sys.executable, os.path.join(GEN, "attn_asm_gen.py"), "--knobs-header", str(h)], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and '#define LL_ASM_KNOBS ""' in h.read_text()
    r = _run([os.path.join(GEN, "attn_asm_gen.py"), "--knobs-header", str(h)], {"ASM_G_NO_X": "1"})
    assert r.returncode != 0                                        # a timing-only knob cannot be compiled into a library without --diag
