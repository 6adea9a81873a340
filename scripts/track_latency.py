#!/usr/bin/env python3
"""Latency of the drop-in Tracker.on_track (batch 1, pose feedback) on one MI355X with a stub
renderer (pre-rendered arrays): what BASELINE config 3 would report as Hz if YCB-Video and a
renderer were available.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import se3tracknet_amd as se3
from oracle import fixtures as Fx
from oracle import se3_oracle as O


class StubRenderer:
    def __init__(self):
        self.rgb, self.depth = Fx.synthetic_render(1, 0.8)

    def render(self, ob2cam, K, window):
        return self.rgb, self.depth


def main(frames=300, faces_subdiv=None, f16x3=False):
    mean, std = Fx.mean_std(0)
    sd = {"state_dict": O.make_state_dict(0, head_gain=0.0005)}
    if faces_subdiv is None:
        trk = se3.Tracker(Fx.DATASET_INFO, mean, std, sd, renderer=StubRenderer())
        rdesc = "stub renderer (pre-rendered arrays)"
    else:  # full pipeline: HIP rasteriser on an icosphere with 20 * 4^subdiv faces (YCB scans: ~1e5 faces)
        from oracle import raster_oracle as R
        mesh = R.icosphere(faces_subdiv, 0.06, 0)
        trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=150.0), mean, std, sd)
        trk.renderer = se3.HipRenderer(trk.engine, mesh)
        rdesc = "HIP rasteriser, %d faces, rendered A stays on the device" % len(mesh["faces"])
    if f16x3:
        trk.engine.set_precision(se3._lib.PREC_F16X3)
        rdesc += ", SE3TN_PREC_F16X3"
    rgb, depth = Fx.synthetic_frame(3)
    P = Fx.pose(3)
    for _ in range(20):
        P = trk.on_track(P, rgb, depth)
    P = Fx.pose(3)
    torch.cuda.synchronize()
    lat = []
    for _ in range(frames):
        t0 = time.perf_counter()
        P = trk.on_track(P, rgb, depth)
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    # device-only time of one batch-1 infer (HIP events inside the library)
    trk.engine.profile_enable(1)
    trk.on_track(P, rgb, depth)
    conv_ms, _, tot_ms = trk.engine.profile_read(0)
    print(json.dumps({"on_track_ms_median": round(float(np.median(lat)), 4), "on_track_ms_p95": round(float(np.percentile(lat, 95)), 4),
                      "hz_median": round(1000.0 / float(np.median(lat)), 1), "device_infer_ms": round(tot_ms, 4),
                      "device_conv_ms": round(conv_ms, 4), "frames": frames,
                      "note": "batch 1, 480x640 frame uploaded per call (pageable H2D), %s, pose D2H sync per frame" % rdesc}))


if __name__ == "__main__":
    main()
    main(faces_subdiv=6)
    main(faces_subdiv=6, f16x3=True)
