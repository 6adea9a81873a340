"""Diagnostic (variant build -DSE3TN_WG_TRACE): where and when every workgroup of the LAST wino_gemm_kernel launch of a batch-64
step ran.  usage (GPU box): SE3TN_LIB=variants/lib_trace.so python scripts/wg_trace.py"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import se3tracknet_amd as se3
from oracle import se3_oracle as O, fixtures as Fx

lib = se3._lib.load()
lib.se3tn_debug_wg_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
eng = se3.Engine(0, 64)
eng.load_state_dict(O.make_state_dict(0))
A, B = Fx.net_inputs(1, 64)
A, B = A.cuda(), B.cuda()
tr = torch.empty((64, 3), device="cuda"); ro = torch.empty((64, 3), device="cuda")
for _ in range(3):
    eng.infer(A, B, 64, se3.NCHW, tr, ro)
torch.cuda.synchronize()
lib.se3tn_debug_wg_trace(None, 0, 1)
eng.infer(A, B, 64, se3.NCHW, tr, ro)
torch.cuda.synchronize()
buf = np.zeros(8 * 4096, np.uint64)
assert lib.se3tn_debug_wg_trace(buf.ctypes.data, buf.nbytes, 0) == 0
t = buf.reshape(4096, 8)
t = t[t[:, 0] > 0]
print("workgroups recorded (last wino_gemm launch wins per id):", len(t))
hw, xcc = t[:, 1].astype(np.int64), t[:, 2].astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 20 + cu
t0 = t[:, 3].min()
st, kd, en = (t[:, 3] - t0) / 100.0, (t[:, 4] - t0) / 100.0, (t[:, 5] - t0) / 100.0     # us (100 MHz)
print("distinct (xcc,se,sh,cu):", len(np.unique(key)), " launch span %.1f us" % en.max())
ids = t[:, 0].astype(np.int64) - 1
for k in np.unique(key)[:3]:
    m = key == k
    o = np.argsort(st[m])
    print("CU", k, [(int(ids[m][i]), round(float(st[m][i]), 1), round(float(kd[m][i]), 1), round(float(en[m][i]), 1)) for i in o])
# first-wave co-residents: which ids share a CU among those that start within 2 us
first = st < 2.0
print("first wave:", first.sum(), "ids range", ids[first].min(), ids[first].max())
pairs = {}
for k, i in zip(key[first], ids[first]):
    pairs.setdefault(k, []).append(int(i))
ex = list(pairs.items())[:6]
print("co-resident first-wave ids (sample):", ex)
d = [abs(v[0] - v[1]) for v in pairs.values() if len(v) == 2]
print("id distance of co-resident pairs: ", np.unique(d, return_counts=True))
dur = en - st
print("tile duration us: median %.1f  p10 %.1f p90 %.1f ; k-loop share %.2f" % (np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), np.median((kd - st) / dur)))
# lock-step measure: for each CU, fraction of time exactly two WGs are in their K-loop / one / none
tot2 = tot1 = tot0 = 0.0
for k in np.unique(key):
    m = key == k
    ev = sorted([(s_, 1) for s_ in st[m]] + [(e_, -1) for e_ in kd[m]])
    cur, last = 0, 0.0
    for x, dlt in ev:
        if cur >= 2: tot2 += x - last
        elif cur == 1: tot1 += x - last
        else: tot0 += x - last
        cur += dlt; last = x
    tot0 += en.max() - last
n = len(np.unique(key))
print("per CU avg us with 2 / 1 / 0 workgroups inside their K-loop: %.1f / %.1f / %.1f" % (tot2 / n, tot1 / n, tot0 / n))
