"""Diagnostic (variant build -DSE3TN_GEMMP_TRACE): per-K-step timeline of the eight waves of workgroup 0 of the LAST wino_gemmp_kernel
launch of a batch-64 step (the heads' second GEMM, K = 512: 32 K-steps over two tiles).
usage (GPU box): SE3TN_LIB=variants/lib_gptrace.so python scripts/gemmp_trace.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import se3tracknet_amd as se3
from oracle import se3_oracle as O, fixtures as Fx

lib = se3._lib.load()
lib.se3tn_debug_gemmp_trace.argtypes = [C.c_void_p, C.c_size_t]
eng = se3.Engine(0, 64)
eng.load_state_dict(O.make_state_dict(0))
A, B = Fx.net_inputs(1, 64)
A, B = A.cuda(), B.cuda()
tr = torch.empty((64, 3), device="cuda"); ro = torch.empty((64, 3), device="cuda")
for _ in range(4):
    eng.infer(A, B, 64, se3.NCHW, tr, ro)
torch.cuda.synchronize()
buf = np.zeros(8 * 64 * 5, np.uint64)
assert lib.se3tn_debug_gemmp_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(8, 64, 5).astype(np.int64)
n = int((t[0, :, 0] > 0).sum())
t = t[:, :n]
t0 = t[:, 0, 0].min()
print("K-steps recorded:", n, " total cycles (wave 0): %d" % (t[0, n - 1, 4] - t[0, 0, 0]))
step = np.diff(t[:, :, 0], axis=1)                # step-to-step period per wave
print("K-step period, cycles: median %d  p10 %d  p90 %d   (ideal 2 waves x 64 MFMAs x 64 clk = 8192)" % (
    np.median(step), np.percentile(step, 10), np.percentile(step, 90)))
seg = np.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2], t[:, :, 4] - t[:, :, 3]], -1)
names = ["DMA issue", "fragment reads + 64 MFMAs issued", "s_waitcnt vmcnt(0)", "s_barrier"]
for i, nm in enumerate(names):
    print("%-36s median %6d  p10 %6d  p90 %6d cycles" % (nm, np.median(seg[..., i]), np.percentile(seg[..., i], 10), np.percentile(seg[..., i], 90)))
print("per wave medians [dma, mfma, vmcnt, barrier]:")
for w in range(8):
    print("  wave %d:" % w, [int(np.median(seg[w, :, i])) for i in range(4)], " start skew vs wave 0 (median): %d" % int(np.median(t[w, :, 0] - t[0, :, 0])))
