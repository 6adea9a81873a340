"""BASELINE configs[0], with the reference's own class: tests/golden/predict_tracker.npz holds 6 frames of the UNMODIFIED
`predict.Tracker` (predict.py:127-296: __init__, render_window, on_track) executed end to end in the build container by
oracle/make_predict_golden.py -- torch-CPU network, the reference's VispyRenderer on a real OpenGL implementation (SwiftShader),
the reference's Utils / datasets / data_augmentation; only cv2.resize(NEAREST) / cv2.Rodrigues are the restated rules (OpenCV absent).
CPU: the oracle's composition of the inner functions reproduces the class.  GPU: the drop-in Tracker, fed the same image A,
reproduces it within the north-star tolerances; with its own HIP rasteriser it stays within what two GL implementations differ by."""
import os

import numpy as np
import pytest

from oracle import fixtures as Fx
from oracle import se3_oracle as O
from oracle.make_predict_golden import FRAMES, HEAD_GAIN, MESH, OBJECT_WIDTH


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "predict_tracker.npz"))


def _inputs(golden, f):
    P = golden["pose0"] if f == 0 else golden["poses"][f - 1]
    rgb, depth = Fx.structured_frame(700 + f)
    return P, rgb, depth


def test_oracle_composition_equals_predict_tracker(golden):
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    mean, std = Fx.mean_std(0)
    assert golden["frame_cnt"] == FRAMES and float(golden["object_width"]) == OBJECT_WIDTH and np.array_equal(golden["K"], Fx.K_YCB)
    moved = 0.0
    for f in range(FRAMES):
        P, rgb, depth = _inputs(golden, f)
        want = golden["poses"][f]
        got, aux = O.on_track(sd, P, rgb, depth, golden["rgbA"][f], golden["depthA"][f], Fx.K_YCB, OBJECT_WIDTH, mean, std,
                              offset_rule="numpy2")   # the golden was made by the reference under NumPy 2 (this image)
        assert np.abs(got - want).max() < 1e-6, (f, np.abs(got - want).max())     # torch-CPU run-to-run / batch-1 noise floor 1e-7
        moved = max(moved, float(np.abs(want - P).max()))
        assert (golden["depthA"][f] > 0).sum() > 2000 and golden["rgbA"][f].dtype == np.uint8
    assert moved > 1e-3                                                           # the poses really change frame to frame
    # the image A the reference rendered (real GL) against the numpy restatement of the pipeline, frame 0
    from oracle import raster_oracle as R
    m = Fx.icosphere(*MESH)
    bb = O.compute_bbox(golden["pose0"], Fx.K_YCB, OBJECT_WIDTH, (1000, -1000, 1000))
    win = (int(bb[:, 1].min()), int(bb[:, 0].min()), int(bb[:, 1].max()), int(bb[:, 0].max()))
    rgb, depth = R.render(m["vertices"], m["normals"].astype(np.float32), (m["colors"] / 255.0).astype(np.float32), m["faces"],
                          golden["pose0"], Fx.K_YCB, win)
    assert ((depth > 0) != (golden["depthA"][0] > 0)).sum() <= 12
    both = (depth > 0) & (golden["depthA"][0] > 0)
    assert np.abs(depth[both].astype(int) - golden["depthA"][0][both].astype(int)).max() <= 1


def test_golden_is_what_predict_tracker_computes_today(golden, tmp_path):
    from oracle import ref_shims, swiftshader_gl as SG
    if not (ref_shims.reference_available() and SG.available()):
        pytest.skip("needs /root/reference and the kaleido wheel's SwiftShader")
    from oracle import make_predict_golden as M
    g = M.run(str(tmp_path))
    assert np.array_equal(g["rgbA"], golden["rgbA"]) and np.array_equal(g["depthA"], golden["depthA"])
    assert np.abs(g["poses"] - golden["poses"]).max() < 1e-6


@pytest.mark.gpu
def test_dropin_tracker_vs_predict_tracker(golden):
    import se3tracknet_amd as se3
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    mean, std = Fx.mean_std(0)
    frame = [0]

    class ReferenceImageA:          # injected renderer: the image A predict.Tracker's own renderer produced for this frame
        def render(self, ob2cam, K, window):
            return golden["rgbA"][frame[0]], golden["depthA"][frame[0]]
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd}, renderer=ReferenceImageA())
    hip = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd})
    for t in (trk, hip):
        t.engine.set_offset_rule("numpy2")     # like for like: the golden is predict.Tracker under NumPy 2 (this image's interpreter)
    hip.renderer = se3.HipRenderer(hip.engine, Fx.icosphere(*MESH))
    worst = worst_hip = 0.0
    for f in range(FRAMES):
        frame[0] = f
        P, rgb, depth = _inputs(golden, f)
        got = trk.on_track(P, rgb, depth, gt_A_in_cam=np.eye(4), gt_B_in_cam=np.eye(4), debug=False, samples=1)
        worst = max(worst, float(np.abs(got - golden["poses"][f]).max()))
        worst_hip = max(worst_hip, float(np.abs(hip.on_track(P, rgb, depth) - golden["poses"][f]).max()))
        # and the rendered image itself: HIP rasteriser vs the reference's renderer on real GL
        rgbA, depthA = hip.render_window(P)
        cov = (depthA > 0) != (golden["depthA"][f] > 0)
        assert cov.sum() <= 20                                   # of 30,976 pixels (silhouette pixels, sub-pixel snapping of the GL implementation)
    print("drop-in Tracker vs predict.Tracker: max |d pose| %.2e (same image A), %.2e (HIP rasteriser's image A)" % (worst, worst_hip))
    assert worst < 1e-5 and trk.frame_cnt == int(golden["frame_cnt"])
    assert worst_hip < 5e-4
