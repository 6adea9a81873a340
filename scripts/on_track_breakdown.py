"""Run on the GPU box: where the wall time of one Tracker.on_track call (batch 1) goes.  Each phase is timed with a device
synchronisation after it (so overlap between phases is removed: the sum exceeds the real per-frame time)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O

mean, std = Fx.mean_std(0)
sd = {"state_dict": O.make_state_dict(0, head_gain=0.0005)}
trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=150.0), mean, std, sd)
trk.renderer = se3.HipRenderer(trk.engine, Fx.icosphere(6, 0.06, 0))
rgb, depth = Fx.synthetic_frame(3)
P = Fx.pose(3)
U = se3.utils
for _ in range(30):
    trk.on_track(P, rgb, depth)
def T(fn, reps=300):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e6 * float(np.median(ts))
dev = "cuda:0"
out = {}
out["whole on_track"] = T(lambda: trk.on_track(P, rgb, depth))
out["compute_bbox x2 + gl_window (host)"] = T(lambda: (U.compute_bbox(P, trk.K, trk.object_width, scale=(1000, 1000, 1000)), se3.HipRenderer.gl_window(P, trk.K, trk.object_width)))
out["H2D rgb+depth frame (pageable)"] = T(lambda: (torch.from_numpy(rgb).to(dev, non_blocking=True), torch.from_numpy(depth.view(np.int16)).to(dev, non_blocking=True)))
win = se3.HipRenderer.gl_window(P, trk.K, trk.object_width)
out["render_device"] = T(lambda: trk.renderer.render_device(P, trk.K, win))
rgb_d = torch.from_numpy(rgb).to(dev); dep_d = torch.from_numpy(depth.view(np.int16)).to(dev)
bb = U.compute_bbox(P, trk.K, trk.object_width, scale=(1000, 1000, 1000))
cropB = dict(rgb=rgb_d, depth=dep_d, window=U.crop_window(bb), z_offset_mm=800.0, stats=1)
out["preprocess (one crop)"] = T(lambda: trk.engine.preprocess([cropB], trk.engine.input_buffer_ptr(1)))
pA = torch.from_numpy(P.reshape(1, 16)).to(dev)
trk._poseA[:1].copy_(pA)
out["infer (n=1)"] = T(lambda: trk.engine.infer(trk.engine.input_buffer_ptr(0), trk.engine.input_buffer_ptr(1), 1, se3.NHWC, trk._trans, trk._rot, trk._poseA, trk._poseB))
out["poseA upload (16 doubles)"] = T(lambda: trk._poseA[:1].copy_(torch.from_numpy(np.tile(P.reshape(1, 16), (1, 1))), non_blocking=True))
out["read back (one D2H + sync)"] = T(lambda: trk._read_back(1))
out["empty sync pair"] = T(lambda: None)
for k, v in out.items():
    print("%-40s %8.1f us" % (k, v))
