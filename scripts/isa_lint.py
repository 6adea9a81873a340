#!/usr/bin/env python3
"""ISA lint of the built library (profiles/EXPERIMENTS.md items 13).

On gfx950 a packed-float32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 ...) whose `op_sel` modifier selects
the HIGH half of src1 for the low result (op_sel:[x,1] / op_sel:[x,1,x]) returns wrong values in lanes 48-63 while waves of
another kernel on the same CU issue the 8-element 16-bit matrix instructions (v_mfma_f32_32x32x16_f16, _16x16x32_f16,
_32x32x16_bf16) -- measured with scripts/probes/pk_opsel.hip.  hipcc emits that form whenever its register allocation leaves a
float pair swapped; this script disassembles every gfx950 code object inside the shared library and lists the kernels that
contain it (src2 selections are flagged too: untested, treated as unsafe).  Exit status 1 if any.

    python scripts/isa_lint.py [path/to/libse3tracknet.so]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\b(v_pk_\w+_f32|v_pk_mov_b32)\b.*\bop_sel:\[([01](?:,[01])+)\]")


def code_objects(lib, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    for k, s in enumerate(starts):
        part = os.path.join(tmp, "bundle%d.bin" % k)
        open(part, "wb").write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        out = os.path.join(tmp, "dev%d.co" % k)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True)
        if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) > 0:
            yield out


def lint(lib):
    """-> (kernels scanned, {kernel: [offending instruction, ...]})"""
    bad, kernels = {}, 0
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1)
                    kernels += 1
                    continue
                m = PK.search(line)
                if m and cur:
                    sel = m.group(2).split(",")
                    if any(b == "1" for b in sel[1:]):           # src1 (or src2) low-half select
                        ins = line.split("//")[0].strip()
                        bad.setdefault(cur, []).append(re.sub(r"\s+", " ", ins))
    return kernels, bad


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "iros20-6d-pose-tracking_amd", "libse3tracknet.so")
    n, bad = lint(lib)
    print("%s: %d kernels scanned, %d with a src1/src2 op_sel on a packed-f32 instruction" % (os.path.basename(lib), n, len(bad)))
    for k, v in sorted(bad.items()):
        print("  %s: %d, e.g. %s" % (k, len(v), v[0]))
    sys.exit(1 if bad else 0)
