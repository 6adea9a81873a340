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
lib.probe_launch2.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
side = torch.zeros(64 * 12, device="cuda")
lib.probe_tail.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]

n = 64
sd = O.make_state_dict(0)
# FC weights in the layout tail_kernel reads them: [2 heads][3][512] and a bias padded to [2][4]
fc_w = torch.cat([sd["trans_out.0.weight"], sd["rot_out.0.weight"]], 0).float().cuda().contiguous()
fc_b = torch.zeros(8, device="cuda"); fc_b[0:3] = sd["trans_out.0.bias"].cuda(); fc_b[4:7] = sd["rot_out.0.bias"].cuda()
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
READERS = [(0, "counted vmcnt, 16-byte loads"), (1, "drained (vmcnt(0)) batches"),
           (3, "counted + 72 bytes of LDS"), (4, "counted + FC arithmetic, no LDS"),
           (5, "counted, v_pk_add_f32 plain (asm)"), (6, "counted, v_pk_add_f32 op_sel swizzle"),
           (7, "DRAINED, v_pk_add_f32 op_sel swizzle"), (8, "no loads: 4096 plain v_pk_add_f32"), (9, "no loads: 4096 swizzled v_pk_add_f32"),
           (11, "copy of the tail, drained"), (10, "copy of the old tail_kernel"), (12, "copy of the old tail + extra stores")]
R = 24
for mode_a in ("direct",):
    ea = mk(mode_a)
    ea.infer(Ac, Bc, n, se3.NCHW, t, r); torch.cuda.synchronize()
    head = ea.debug_buffer("head", n).reshape(n, 169, 1024)
    want = torch.zeros((n, 1024), device="cuda")
    for p in range(169):
        want = want + head[:, p, :]
    ref_logits = ea.logits(n).clone()      # the shipped (drained) tail's logits of the same tensor
    ptr = C.c_void_p(); dims = (C.c_int32 * 3)()
    assert ea.lib.se3tn_debug_buffer(ea._h, b"head", C.byref(ptr), dims) == 0
    for mode_b in ("f16x3",):
        eb = mk(mode_b) if mode_b else None
        for which, rname in READERS:
            outs = [torch.empty((n, 1024), device="cuda") for _ in range(R)]
            pooled_o, wave_o = [], []
            torch.cuda.synchronize()
            for i in range(R):
                if eb is not None:
                    with torch.cuda.stream(s2):
                        eb.infer(Ac, Bc, n, se3.NCHW, t2, r2)
                with torch.cuda.stream(s1):
                    ea.infer(Ac, Bc, n, se3.NCHW, t, r)
                    if which >= 10:
                        pooled_o.append(torch.empty((n, 1024), device="cuda")); wave_o.append(torch.empty((n, 12), device="cuda"))
                        assert lib.probe_tail(which - 10, ptr, fc_w.data_ptr(), fc_b.data_ptr(), outs[i].data_ptr(),
                                              pooled_o[-1].data_ptr(), wave_o[-1].data_ptr(), n, C.c_void_p(s1.cuda_stream)) == 0
                    elif which in (3, 4, 5, 6, 7, 8, 9):
                        assert lib.probe_launch2(which, ptr, outs[i].data_ptr(), fc_w.data_ptr(), side.data_ptr(), n,
                                                 C.c_void_p(s1.cuda_stream)) == 0
                    else:
                        assert lib.probe_launch(which, ptr, outs[i].data_ptr(), n, C.c_void_p(s1.cuda_stream)) == 0
            torch.cuda.synchronize()
            if which in (8, 9):
                tt = torch.arange(256, device="cuda", dtype=torch.float32)
                vw = torch.zeros((n, 1024), device="cuda")
                vw[:, 0::4] = 4096.0 * (tt + 1); vw[:, 1::4] = 4096.0 * (2 * tt + 1)
            w_ = ref_logits if which >= 10 else (vw if which in (8, 9) else want)
            cmp = [(o.flatten()[:n * 6].reshape(n, 6) if which >= 10 else o) for o in outs]
            bad_l = sum(1 for o in cmp if not torch.equal(o, w_))
            bad_e = sum(int((o != w_).sum()) for o in cmp)
            print("stream 1: %-6s infer + reader %-30s | stream 2: %-6s : %2d / %d reader launches differ (%d elements)" % (
                mode_a, rname, mode_b or "idle", bad_l, R, bad_e))
