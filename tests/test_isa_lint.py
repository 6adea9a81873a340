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


def _chains_module():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    spec = importlib.util.spec_from_file_location("isa_chains", os.path.join(ROOT, "scripts", "isa_chains.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_chain_figures_of_a_sequence():
    m = _chains_module()
    # 3 reads waited together, then `read -> wait -> store` four times, then a wait that only has a store before it
    seq = ["L", "L", "L", "w0", "b", "L", "w0", "S", "b", "L", "w0", "S", "b", "L", "w0", "S", "L", "w1", "w0", "S", "w0"]
    assert m.figures(seq) == (5, 3, 1)
    assert m.figures(["D"] * 10 + ["w0", "|"]) == (1, 10, 0)
    assert m.figures([]) == (0, 0, 0)


def test_streaming_passes_of_the_float32_batch_path_keep_their_reads_in_flight():
    """profiles/EXPERIMENTS.md item 38: the F(6x6) out-transform and tail issued one residual read per wait (36 dependent memory
    latencies per thread) until the reads were hoisted; a refactoring that puts a read back inside the per-pixel branch shows up
    here, in the disassembly, not in any parity test."""
    m = _chains_module()
    if not (os.path.exists(os.path.join(m.LLVM, "llvm-objdump")) and os.path.exists(LIB)):
        pytest.skip("needs the ROCm LLVM tools and the built library")
    seqs = m.sequences(LIB)
    names = m.demangle(list(seqs))
    by_name = {names[k].replace("se3tn::", "").replace("void ", "").split("(")[0]: m.figures(v) for k, v in seqs.items()}
    want = {  # kernel: (most chains, fewest reads in flight at the widest point)
        "wino_input_kernel<6, 2, 0>": (2, 60),
        "wino_mid_kernel<6, 4, 32, 1, 0>": (2, 60),
        "wino_mid_kernel<6, 2, 64, 1, 0>": (2, 60),
        "wino_output_kernel<6, 2, 1, 0>": (6, 36),
        "wino_tail_kernel<6, 2, 64, 0>": (6, 36),
        "wino_output_kernel<4, 2, 1, 0>": (6, 16),
        "wino_tail_kernel<4, 3, 64, 0>": (6, 16),
        "maxpool3x3s2_kernel": (4, 16),
    }
    missing = [k for k in want if k not in by_name]
    assert not missing, "kernels not found in the library: %s (have e.g. %s)" % (missing, sorted(by_name)[:5])
    for k, (max_chains, min_inflight) in want.items():
        chains, inflight, _ = by_name[k]
        assert chains <= max_chains and inflight >= min_inflight, (k, chains, inflight)
