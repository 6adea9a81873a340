"""Driver of scripts/probes/vmcnt_race.hip (run on the GPU box; see that file).  The reader kernels sum a constant
buffer on stream 1 while stream 2 runs the library (one Engine) in a chosen mode, or nothing.  Prints, per
(reader, co-runner), in how many launches and elements the sums differ from the sequentially rounded reference."""
import ctypes
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import se3tracknet_amd as se3
from oracle import fixtures as Fx
from oracle import se3_oracle as O

lib = ctypes.CDLL(os.path.abspath("variants/libvmcnt_probe.so"))
lib.probe_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
lib.probe_write.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
FRESH = os.environ.get("FRESH", "1") == "1"     # 1: a writer kernel re-writes the buffer (same values) before every reader
n = 64
torch.manual_seed(0)
x = torch.randn((n, 169, 1024), device="cuda")
master = x.clone()
want = torch.zeros((n, 1024), device="cuda")
for p in range(169):
    want = want + x[:, p, :]                      # same order, one rounding per add, as the kernels
sd = O.make_state_dict(0)
A, B = Fx.net_inputs(5, n)
Ac, Bc = A.cuda(), B.cuda()
eng = se3.Engine(0, n)
eng.load_state_dict(sd)
t = torch.empty((n, 3), device="cuda"); r = torch.empty((n, 3), device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
READERS = [(0, "counted vmcnt, 16-byte loads"), (2, "counted vmcnt, 8-byte loads"), (1, "drained (vmcnt(0)) batches")]
MODES = [("nothing", None), ("library f32 (Winograd)", "f32"), ("library f32 direct", "direct"), ("library f16x3", "f16x3")]
R = 60
print('buffer re-written by a writer kernel before every reader:', FRESH)
for mode_name, mode in MODES:
    if mode == "f32":
        eng.set_precision(se3._lib.PREC_F32); eng.set_winograd(6)
    elif mode == "direct":
        eng.set_precision(se3._lib.PREC_F32); eng.set_winograd(0)
    elif mode == "f16x3":
        eng.set_precision(se3._lib.PREC_F16X3)
    for which, rname in READERS:
        outs = [torch.empty((n, 1024), device="cuda") for _ in range(R)]
        torch.cuda.synchronize()
        for i in range(R):
            if mode is not None:
                with torch.cuda.stream(s2):
                    eng.infer(Ac, Bc, n, se3.NCHW, t, r)
            with torch.cuda.stream(s1):
                if FRESH:
                    assert lib.probe_write(master.data_ptr(), x.data_ptr(), n * 169, ctypes.c_void_p(s1.cuda_stream)) == 0
                rc = lib.probe_launch(which, x.data_ptr(), outs[i].data_ptr(), n, ctypes.c_void_p(s1.cuda_stream))
                assert rc == 0
        torch.cuda.synchronize()
        bad_l = 0; bad_e = 0; lanes = np.zeros(64, np.int64); comps = np.zeros(4, np.int64)
        for o in outs:
            ne = (o != want)
            if ne.any():
                bad_l += 1; bad_e += int(ne.sum())
                idx = torch.nonzero(ne)[:, 1].cpu().numpy()          # channel = 4 * thread + component
                np.add.at(lanes, (idx // 4) % 64, 1); np.add.at(comps, idx % 4, 1)
        msg = "co-runner %-24s reader %-30s: %2d / %d launches differ, %6d elements" % (mode_name, rname, bad_l, R, bad_e)
        if bad_e:
            msg += "; by lane quarter %s, by component xyzw %s" % ([int(lanes[q * 16:(q + 1) * 16].sum()) for q in range(4)], comps.tolist())
        print(msg)
