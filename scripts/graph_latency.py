import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O
from oracle import raster_oracle as R
mean, std = Fx.mean_std(0)
sd = {"state_dict": O.make_state_dict(0, head_gain=0.0005)}
mesh = R.icosphere(6, 0.06, 0)
rgb, depth = Fx.synthetic_frame(3)
for ug in (False, True, False, True):
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=150.0), mean, std, sd, use_graphs=ug)
    trk.renderer = se3.HipRenderer(trk.engine, mesh)
    P = Fx.pose(3)
    for _ in range(30): P = trk.on_track(P, rgb, depth)
    P = Fx.pose(3); torch.cuda.synchronize()
    lat = []
    for _ in range(300):
        t0 = time.perf_counter(); P = trk.on_track(P, rgb, depth); lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    print("use_graphs=%s  median %.4f ms  p95 %.4f  -> %.0f Hz" % (ug, np.median(lat), np.percentile(lat, 95), 1000 / np.median(lat)))
