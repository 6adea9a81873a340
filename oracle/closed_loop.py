"""Closed-loop stand-in for BASELINE configs[2] (YCB-Video seq 0048 full track: data and pretrained weights are
not available offline): a synthetic sequence driven through the drop-in ``Tracker.on_track`` with pose
feedback, the HIP rasteriser producing image A every frame, checked frame by frame against the CPU oracle.

TEST INFRASTRUCTURE ONLY (imports the oracle).  Used by tests/test_closed_loop.py and, as the checker, by
bench.py's `track` block; never by the product package.

Per frame f with the tracker's own previous pose P_f (teacher-forced: errors cannot compound chaotically and
every frame is a full-strength check):
    HIP:     P_{f+1} = tracker.on_track(P_f, rgb_f, depth_f)         (render -> crop -> normalise -> CNN -> pose)
    oracle:  O.on_track(sd, P_f, rgb_f, depth_f, rgbA_f, depthA_f)   fed the SAME rendered image A (read back)
    checks:  integer bbox identical (the only discrete decisions on the path),
             |d(trans, rot)| <= 1e-4, |d pose| <= 1e-5  (the north-star tolerances).
The loop of predict.py:529-564: prev_pose <- cur_pose, no re-initialisation -- except the safety net below."""
import time

import numpy as np

from . import fixtures as Fx
from . import se3_oracle as O

OBJECT_WIDTH_MM = 150.0
HEAD_GAIN = 0.00002      # random-init FC gain: ~0.5 mm / 0.08 deg of pose change per frame, 300 frames stay in the frustum
N_DISTINCT_FRAMES = 16   # the observed frames cycle through this many synthetic 480x640 RGB-D images


def _lost(P):
    """A random-init network has no reason to stay on the object: if the pose drifts out of the region where the
    crop window is well defined, the driver re-detects (resets to the initial pose), as a real system would."""
    t = P[:3, 3]
    return not (abs(t[0]) < 0.25 and abs(t[1]) < 0.2 and 0.45 < t[2] < 1.3)


def make_tracker(se3, subdiv=5, precision=None):
    from . import raster_oracle as R
    mean, std = Fx.mean_std(0)
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    for k in ("trans_out.0.bias", "rot_out.0.bias"):     # the FC biases alone would move the pose 1.5 mm per frame
        sd[k] = sd[k] * 0.1
    mesh = R.icosphere(subdiv, 0.06, 0)                   # 20 * 4^subdiv faces
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH_MM), mean, std, {"state_dict": sd})
    trk.renderer = se3.HipRenderer(trk.engine, mesh)
    if precision is not None:
        trk.engine.set_precision(precision)
    return trk, sd, (mean, std), len(mesh["faces"])


def run(se3, frames=300, check=True, subdiv=5, precision=None, timing=True):
    """Returns the `track` block of the bench line."""
    trk, sd, (mean, std), nfaces = make_tracker(se3, subdiv, precision)
    seq = [Fx.synthetic_frame(400 + i) for i in range(N_DISTINCT_FRAMES)]
    P0 = Fx.pose(3, (0.02, -0.01, 0.8))
    out = {"frames": frames, "renderer": "HIP rasteriser, %d-face vertex-colour mesh, image A stays on the device" % nfaces,
           "sequence": "synthetic 480x640 RGB-D frames (%d distinct, cycled), random-init weights, pose feedback "
                       "frame to frame" % N_DISTINCT_FRAMES}
    poses_timed = None
    if timing:
        P = P0.copy()
        for f in range(10):                                   # warm-up
            trk.on_track(P, *seq[f % N_DISTINCT_FRAMES])
        P = P0.copy()
        lat, poses_timed, reinits = [], [], 0
        for f in range(frames):
            rgb, depth = seq[f % N_DISTINCT_FRAMES]
            t0 = time.perf_counter()
            P = trk.on_track(P, rgb, depth)
            lat.append(time.perf_counter() - t0)
            poses_timed.append(P)
            if _lost(P):
                P = P0.copy(); reinits += 1
        lat = np.array(lat) * 1e3
        out.update(hz=round(1000.0 / float(np.median(lat)), 1), ms_median=round(float(np.median(lat)), 4),
                   ms_p95=round(float(np.percentile(lat, 95)), 4), reinits=reinits,
                   includes="per frame: pageable H2D of the 480x640 frame, render, crop+normalise, CNN, pose update, "
                            "D2H of the pose (one sync), as predict.py:217-296 without its GUI / second render")
    if check:
        P = P0.copy()
        bbox_mismatch = 0
        e_net = e_pose = 0.0
        drift = 0.0
        replay_diff = 0.0
        reinits_c = 0
        for f in range(frames):
            rgb, depth = seq[f % N_DISTINCT_FRAMES]
            Pn = trk.on_track(P, rgb, depth)
            rgbA = trk.renderer.rgb.cpu().numpy()               # the image A this frame was computed from
            depthA = trk.renderer.depth.cpu().numpy().view(np.uint16)
            want, aux = O.on_track(sd, P, rgb, depth, rgbA, depthA, trk.K, trk.object_width, mean, std,
                                   trk.trans_normalizer, trk.rot_normalizer)
            bbox_mismatch += int(not np.array_equal(trk.last_prediction["bbox"], aux["bbox"]))
            e_net = max(e_net, float(np.abs(trk.last_prediction["trans"][0] - aux["trans"]).max()),
                        float(np.abs(trk.last_prediction["rot"][0] - aux["rot"]).max()))
            e_pose = max(e_pose, float(np.abs(Pn - want).max()))
            if poses_timed is not None:
                replay_diff = max(replay_diff, float(np.abs(Pn - poses_timed[f]).max()))
            drift = max(drift, float(np.linalg.norm(Pn[:3, 3] - P0[:3, 3])))
            if _lost(Pn):
                reinits_c += 1
                P = P0.copy()
            else:
                P = Pn
        out.update(frames_checked=frames, bbox_mismatches=bbox_mismatch, max_abs_trans_rot=e_net, max_abs_pose=e_pose,
                   tol_trans_rot=1e-4, tol_pose=1e-5, max_drift_m=round(drift, 4), reinits_checked_pass=reinits_c,
                   ok=bool(bbox_mismatch == 0 and e_net <= 1e-4 and e_pose <= 1e-5))
        if poses_timed is not None:
            out["timed_vs_checked_pass_max_abs_pose"] = replay_diff   # the two passes are the same deterministic track
    return out
