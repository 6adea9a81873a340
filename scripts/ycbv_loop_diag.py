"""Where does the closed-loop getResultsYcb run leave the reference's?  Per frame: open-loop error (the reference run's own input pose
fed) and closed-loop error, for {one call, step by step} x {batch-1 kernels on, off}."""
import glob, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O, ycbv_fixtures as YF
from oracle.make_gl_golden import write_ply
from oracle.make_predict_golden import HEAD_GAIN, MESH, OBJECT_WIDTH

g = np.load("tests/golden/driver_ycbv.npz")
tmp = tempfile.mkdtemp()
tree = YF.make_tree(tmp)
sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
mean, std = Fx.mean_std(0)
ply = os.path.join(tmp, "model.ply")
write_ply(ply, Fx.icosphere(*MESH))
frames = []
for s, n in ((48, 9), (50, 6)):
    d = os.path.join(tree, "data_organized", "%04d" % s)
    rf = sorted(glob.glob(os.path.join(d, "color", "*"))); df = sorted(glob.glob(os.path.join(d, "depth_filled", "*")))
    for i in range(1, n):
        frames.append((se3.sequence.read_rgb(rf[i]), se3.sequence.read_depth_mm(df[i])))
want = np.concatenate([g["res_poses"][1:9], g["res_poses"][10:]])
for one_call in (True, False):
    for small in (True, False):
        trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd}, model_path=ply)
        trk.engine.set_offset_rule("numpy2")
        trk.engine.set_small_kernels(small)
        trk.one_call = one_call
        ol, cl, same = [], [], []
        for k in range(13):
            out = trk.on_track(g["res_poses_in"][k], *frames[k])
            ol.append(np.abs(out - want[k]).max())
            a = trk.renderer.rgb.cpu().numpy(); b = trk.renderer.depth.cpu().numpy().view(np.uint16)
            assert np.array_equal(a, g["res_rgbA"][k]) and np.array_equal(b, g["res_depthA"][k])
        prev = None
        for k in range(13):
            if k in (0, 8): prev = g["res_poses_in"][k]
            prev = trk.on_track(prev, *frames[k])
            cl.append(np.abs(prev - want[k]).max())
            a = trk.renderer.rgb.cpu().numpy(); b = trk.renderer.depth.cpu().numpy().view(np.uint16)
            same.append(int((a != g["res_rgbA"][k]).any(axis=2).sum() + 0), )
        print("one_call=%d small=%d" % (one_call, small))
        print("  open loop  :", " ".join("%.1e" % v for v in ol))
        print("  closed loop:", " ".join("%.1e" % v for v in cl))
        print("  image A px differing (closed loop):", same)
