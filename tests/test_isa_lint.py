"""The built library must not contain the packed-float32 form that gfx950 mis-executes beside 16-bit MFMAs of another
kernel (profiles/EXPERIMENTS.md items 13; scripts/isa_lint.py disassembles every code object in the .so)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "iros20-6d-pose-tracking_amd", "libse3tracknet.so")


def _lint_module():
    spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "scripts", "isa_lint.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pattern_flags_only_src1_src2_selects():
    m = _lint_module()

    def flagged(line):
        g = m.PK.search(line)
        return bool(g and any(b == "1" for b in g.group(2).split(",")[1:]))
    assert flagged("v_pk_add_f32 v[4:5], v[4:5], v[8:9] op_sel:[0,1] op_sel_hi:[1,0]")
    assert flagged("v_pk_fma_f32 v[12:13], s[22:23], v[2:3], v[12:13] op_sel:[0,1,0] op_sel_hi:[1,0,1]")
    assert flagged("v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]")
    assert not flagged("v_pk_add_f32 v[4:5], v[4:5], v[8:9]")
    assert not flagged("v_pk_mul_f32 v[2:3], v[8:9], v[2:3] op_sel:[1,0] op_sel_hi:[0,1]")      # src0 only: measured safe
    assert not flagged("v_pk_fma_f32 v[2:3], v[18:19], v[10:11], v[2:3] op_sel_hi:[0,1,1]")     # op_sel_hi only: measured safe
    assert not flagged("v_pk_fma_f16 v2, v3, v4, v2 op_sel:[0,1,0]")                             # not a float32 instruction


def test_built_library_is_clean():
    m = _lint_module()
    if not (os.path.exists(os.path.join(m.LLVM, "llvm-objdump")) and os.path.exists(LIB)):
        pytest.skip("needs the ROCm LLVM tools and the built library")
    kernels, bad = m.lint(LIB)
    assert kernels >= 60, "the disassembly found only %d kernels: the lint is not looking at the library" % kernels
    assert not bad, "kernels with a src1/src2 op_sel on a packed-f32 instruction: %s" % {k: v[0] for k, v in bad.items()}
