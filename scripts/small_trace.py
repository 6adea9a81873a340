"""Phase timeline inside conv_slices_small_kernel (variant built with -DSE3TN_SMALL_TRACE): per workgroup, 100 MHz stamps at
entry | DMA issued | first data landed | K-step 3 | K-step 8 | accumulators final | partial sums stored."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O
m = se3.Se3TrackNet(176, max_batch=1); m.load_state_dict(O.make_state_dict(0)); m.cuda(0)
A, B = Fx.net_inputs(1, 1); Ac, Bc = A.cuda(), B.cuda()
for _ in range(20): m(Ac, Bc, return_feature=False)
torch.cuda.synchronize()
lib = m.engine.lib
buf2 = np.zeros((2, 1024, 8), np.uint64)
lib.se3tn_debug_trace_slices.argtypes = [C.c_void_p]
assert lib.se3tn_debug_trace_slices(buf2.ctypes.data) == 0
buf = buf2[0]
cyc = buf2[1][:256, :7].astype(np.int64)
wal = buf2[0][:256, :7].astype(np.int64)
ratio = (cyc[:, 5] - cyc[:, 2]) / np.maximum(wal[:, 5] - wal[:, 2], 1) * 100.0     # clock64 ticks per microsecond over the K-step phase
print("clock64() ticks per us of wall_clock64() over the K-step phase: mean %.1f [%.1f .. %.1f]  (100 = a constant 100 MHz counter; ~2400 = the core clock)" % (ratio.mean(), ratio.min(), ratio.max()))
t = buf[:256, :7].astype(np.int64)            # the last conv_slices launch: trans|rot conv2.conv2 (256 workgroups)
t0 = t[:, 0].min()
rel = (t - t0) / 100.0
names = ["entry", "DMA issued", "first data", "K-step 3 done", "K-step 8 done", "acc final", "stored"]
print("trans|rot conv2.conv2 (conv_slices_small_kernel<2,1,169>), us after the first workgroup's entry: mean [min .. max] over 256 workgroups")
for i, nme in enumerate(names):
    print("  %-14s %6.2f [%6.2f .. %6.2f]" % (nme, rel[:, i].mean(), rel[:, i].min(), rel[:, i].max()))
d = np.diff(rel, axis=1)
print("  per workgroup: " + " | ".join("%s %.2f" % (n, v) for n, v in zip(["issue", "wait first", "steps 0-3", "steps 4-8", "steps 9-17", "store"], d.mean(0))))

if hasattr(lib, "se3tn_debug_trace_stem"):
    b2 = np.zeros((512, 8), np.uint64)
    lib.se3tn_debug_trace_stem.argtypes = [C.c_void_p]
    assert lib.se3tn_debug_trace_stem(b2.ctypes.data) == 0
    t = b2[:242].astype(np.int64)
    rel = (t - t[:, 0].min()) / 100.0
    names = ["entry", "loads issued", "data landed", "block pair 0", "block pair 1", "block pair 2", "SELU + LDS done", "pooled + stored"]
    print("stem_pool_small_kernel, us after the first workgroup's entry: mean [min .. max] over 242 workgroups")
    for i, nme in enumerate(names):
        print("  %-16s %6.2f [%6.2f .. %6.2f]" % (nme, rel[:, i].mean(), rel[:, i].min(), rel[:, i].max()))
