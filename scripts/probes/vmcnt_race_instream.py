"""Second driver of scripts/probes/vmcnt_race.hip (run on the GPU box): the probe readers take the place of the
library's tail -- stream 1: Engine A infer (f16x3), then a probe reader over A's own head tensor (the float32 output
of the last conv, the buffer tail_kernel reads); stream 2: Engine B infers in the chosen mode.  Reference sums come
from a synchronised copy of the same tensor (identical inputs every launch, so the tensor never changes)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import se3tracknet_amd as se3
from oracle import fixtures as Fx
from oracle import se3_oracle as O

lib = C.CDLL(os.path.abspath("variants/libvmcnt_probe.so"))
lib.probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
n = 64
sd = O.make_state_dict(0)
A, B = Fx.net_inputs(5, n)
Ac, Bc = A.cuda(), B.cuda()
t = torch.empty((n, 3), device="cuda"); r = torch.empty((n, 3), device="cuda")
t2 = torch.empty((n, 3), device="cuda"); r2 = torch.empty((n, 3), device="cuda")


def mk(mode):
    e = se3.Engine(0, n); e.load_state_dict(sd)
    if mode == "f16x3":
        e.set_precision(se3._lib.PREC_F16X3)
    elif mode == "direct":
        e.set_winograd(0)
    return e


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
READERS = [(0, "counted vmcnt, 16-byte loads"), (1, "drained (vmcnt(0)) batches")]
R = 24
for mode_a in ("f16x3", "direct"):
    ea = mk(mode_a)
    ea.infer(Ac, Bc, n, se3.NCHW, t, r); torch.cuda.synchronize()
    head = ea.debug_buffer("head", n).reshape(n, 169, 1024)
    want = torch.zeros((n, 1024), device="cuda")
    for p in range(169):
        want = want + head[:, p, :]
    ptr = C.c_void_p(); dims = (C.c_int32 * 3)()
    assert ea.lib.se3tn_debug_buffer(ea._h, b"head", C.byref(ptr), dims) == 0
    for mode_b in (None, "f32", "f16x3"):
        eb = mk(mode_b) if mode_b else None
        for which, rname in READERS:
            outs = [torch.empty((n, 1024), device="cuda") for _ in range(R)]
            torch.cuda.synchronize()
            for i in range(R):
                if eb is not None:
                    with torch.cuda.stream(s2):
                        eb.infer(Ac, Bc, n, se3.NCHW, t2, r2)
                with torch.cuda.stream(s1):
                    ea.infer(Ac, Bc, n, se3.NCHW, t, r)
                    assert lib.probe_launch(which, ptr, outs[i].data_ptr(), n, C.c_void_p(s1.cuda_stream)) == 0
            torch.cuda.synchronize()
            bad_l = sum(1 for o in outs if not torch.equal(o, want))
            bad_e = sum(int((o != want).sum()) for o in outs)
            print("stream 1: %-6s infer + reader %-30s | stream 2: %-6s : %2d / %d reader launches differ (%d elements)" % (
                mode_a, rname, mode_b or "idle", bad_l, R, bad_e))
