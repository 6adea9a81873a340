"""BASELINE configs[0], with the reference's own class: tests/golden/predict_tracker.npz holds 6 frames of the UNMODIFIED
`predict.Tracker` (predict.py:127-296: __init__, render_window, on_track) executed end to end in the build container by
oracle/make_predict_golden.py -- torch-CPU network, the reference's VispyRenderer on a real OpenGL implementation (SwiftShader),
the reference's Utils / datasets / data_augmentation; only cv2.resize(NEAREST) / cv2.Rodrigues are the restated rules (OpenCV absent).
CPU: the oracle's composition of the inner functions reproduces the class.  GPU: the drop-in Tracker, fed the same image A,
reproduces it within the north-star tolerances -- and so it does END TO END with its own HIP rasteriser (round 5): image A byte-identical
to what the reference's renderer produced on the real GL, poses within 1e-5."""
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


def test_oracle_composition_equals_predict_tracker(golden, tmp_path):
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
    # the image A the reference rendered (real GL) against the statement of that GL's arithmetic, every frame, every byte
    from oracle import ply_io, ss_rules as S
    from oracle.make_gl_golden import write_ply
    ply = os.path.join(str(tmp_path), "model.ply")
    write_ply(ply, Fx.icosphere(*MESH))
    v = ply_io.read_ply(ply)["vertex"]
    nrm = np.stack([v["nx"], v["ny"], v["nz"]], -1)
    nrm = (nrm / np.linalg.norm(nrm, axis=1).reshape(-1, 1)).astype(np.float32)
    col = (np.stack([v["red"], v["green"], v["blue"]], -1) / 255.0).astype(np.float32)
    for f in range(FRAMES):
        P = _inputs(golden, f)[0]
        bb = O.compute_bbox(P, Fx.K_YCB, OBJECT_WIDTH, (1000, -1000, 1000))
        win = (int(bb[:, 1].min()), int(bb[:, 0].min()), int(bb[:, 1].max()), int(bb[:, 0].max()))
        rgb, depth = S.render_vispy(np.stack([v["x"], v["y"], v["z"]], -1), nrm, col, Fx.icosphere(*MESH)["faces"], P, Fx.K_YCB, win)
        assert np.array_equal(rgb, golden["rgbA"][f]) and np.array_equal(depth, golden["depthA"][f]), f


def test_golden_is_what_predict_tracker_computes_today(golden, tmp_path):
    from oracle import ref_shims, swiftshader_gl as SG
    if not (ref_shims.reference_available() and SG.available()):
        pytest.skip("needs /root/reference and the kaleido wheel's SwiftShader")
    from oracle import make_predict_golden as M
    g = M.run(str(tmp_path))
    assert np.array_equal(g["rgbA"], golden["rgbA"]) and np.array_equal(g["depthA"], golden["depthA"])
    assert np.abs(g["poses"] - golden["poses"]).max() < 1e-6


@pytest.mark.gpu
def test_dropin_tracker_vs_predict_tracker(golden, tmp_path):
    import se3tracknet_amd as se3
    from oracle.make_gl_golden import write_ply
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    mean, std = Fx.mean_std(0)
    frame = [0]

    class ReferenceImageA:          # injected renderer: the image A predict.Tracker's own renderer produced for this frame
        def render(self, ob2cam, K, window):
            return golden["rgbA"][frame[0]], golden["depthA"][frame[0]]
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd}, renderer=ReferenceImageA())
    ply = os.path.join(str(tmp_path), "model.ply")
    write_ply(ply, Fx.icosphere(*MESH))                          # the model file predict.Tracker was given
    hip = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd}, model_path=ply)
    assert isinstance(hip.renderer, se3.HipRenderer)
    for t in (trk, hip):
        t.engine.set_offset_rule("numpy2")     # like for like: the golden is predict.Tracker under NumPy 2 (this image's interpreter)
    worst = worst_hip = 0.0
    P_hip = golden["pose0"]
    for f in range(FRAMES):
        frame[0] = f
        P, rgb, depth = _inputs(golden, f)
        got = trk.on_track(P, rgb, depth, gt_A_in_cam=np.eye(4), gt_B_in_cam=np.eye(4), debug=False, samples=1)
        worst = max(worst, float(np.abs(got - golden["poses"][f]).max()))
        # the whole drop-in, its own rasteriser included, in CLOSED LOOP (its own pose fed back, as predict.py's drivers do)
        rgbA, depthA = hip.render_window(P)                      # at the pose predict.Tracker rendered: every byte
        assert np.array_equal(rgbA, golden["rgbA"][f]) and np.array_equal(depthA, golden["depthA"][f]), f
        P_hip = hip.on_track(P_hip, rgb, depth)
        worst_hip = max(worst_hip, float(np.abs(P_hip - golden["poses"][f]).max()))
    print("drop-in Tracker vs predict.Tracker: max |d pose| %.2e (reference's image A injected), %.2e (closed loop, HIP rasteriser's image A: "
          "byte-identical on all %d frames)" % (worst, worst_hip, FRAMES))
    assert worst < 1e-5 and trk.frame_cnt == int(golden["frame_cnt"])
    assert worst_hip < 1e-5
