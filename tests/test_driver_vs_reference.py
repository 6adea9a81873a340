"""The headless YCBInEOAT sequence driver against the reference's own driver.  tests/golden/driver_ycbineoat.npz = the result files
of the UNMODIFIED `predict.predictSequenceYcbInEOAT()` (predict.py:578-626) run end to end in the build container
(oracle/make_driver_golden.py: the reference's Tracker with the 30-degree normaliser, its renderer on a real GL, torch-CPU) on a
synthetic video, plus the image A of every frame.  GPU: sequence.predict_sequence_ycbineoat with the drop-in Tracker on the same
video (regenerated from the fixtures) writes the same files with the same poses."""
import os

import numpy as np
import pytest

from oracle import fixtures as Fx
from oracle import se3_oracle as O
from oracle.make_driver_golden import N_FRAMES, VIDEO, make_video
from oracle.make_predict_golden import HEAD_GAIN, MESH, OBJECT_WIDTH


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "driver_ycbineoat.npz"))


def test_reference_driver_golden_facts(golden, tmp_path):
    assert [str(f) for f in golden["files"]] == ["%07d.txt" % i for i in range(N_FRAMES)]      # frame 0 is tracked too; %07d.txt
    video = make_video(str(tmp_path))
    first = np.loadtxt(os.path.join(video, "annotated_poses", "0000000.txt"))
    assert np.allclose(golden["poses_in"][0], first)                                            # initial pose = annotated_poses[0]
    for i in range(1, N_FRAMES):                                                                # pose feedback, no re-initialisation
        assert np.allclose(golden["poses_in"][i], golden["poses"][i - 1], atol=1e-12)
    # the oracle's composition reproduces the reference driver's poses (30-degree rotation normaliser, predict.py:586)
    from PIL import Image
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    mean, std = Fx.mean_std(0)
    for i in range(N_FRAMES):
        rgb = np.array(Image.open(os.path.join(video, "rgb", "%07d.png" % i)))[:, :, :3]
        depth = np.array(Image.open(os.path.join(video, "depth_filled", "%07d.png" % i))).astype(np.uint16)
        want, _ = O.on_track(sd, golden["poses_in"][i], rgb, depth, golden["rgbA"][i], golden["depthA"][i], Fx.K_YCB, OBJECT_WIDTH,
                             mean, std, 0.03, 30 * np.pi / 180, offset_rule="numpy2")   # the golden: the reference under NumPy 2
        assert np.abs(want - golden["poses"][i]).max() < 1e-6, i


@pytest.mark.gpu
def test_dropin_driver_writes_what_the_reference_driver_writes(golden, tmp_path):
    import se3tracknet_amd as se3
    video = make_video(str(tmp_path))
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    mean, std = Fx.mean_std(0)
    calls = [0]

    class ReferenceImageA:
        def render(self, ob2cam, K, window):
            i = calls[0]; calls[0] += 1
            assert np.abs(np.asarray(ob2cam) - golden["poses_in"][i]).max() < 1e-5       # the pose fed back is the reference's
            return golden["rgbA"][i], golden["depthA"][i]
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd}, renderer=ReferenceImageA(),
                      trans_normalizer=0.03, rot_normalizer=30 * np.pi / 180)
    trk.engine.set_offset_rule("numpy2")       # like for like: the golden is the reference's driver under NumPy 2 (this image's interpreter)
    out = str(tmp_path / "res" / VIDEO)
    res = se3.sequence.predict_sequence_ycbineoat(trk, video, out)
    assert sorted(os.listdir(out)) == [str(f) for f in golden["files"]]
    worst = 0.0
    for i, f in enumerate(golden["files"]):
        worst = max(worst, float(np.abs(np.loadtxt(os.path.join(out, str(f))) - golden["poses"][i]).max()))
    print("drop-in driver vs predict.predictSequenceYcbInEOAT result files: max |d pose| %.2e over %d frames" % (worst, len(golden["files"])))
    assert worst < 1e-5 and res["frames"] == N_FRAMES
    # the same video END TO END: the drop-in loads the model file the reference loaded and renders image A itself (round 5)
    from oracle.make_gl_golden import write_ply
    ply = str(tmp_path / "model.ply")
    write_ply(ply, Fx.icosphere(*MESH))
    hip = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd}, model_path=ply,
                      trans_normalizer=0.03, rot_normalizer=30 * np.pi / 180)
    hip.engine.set_offset_rule("numpy2")
    seen = []
    on_track = hip.on_track

    def recording(*a, **k):                     # image A of every frame: it stays in the renderer's device buffers on both code paths
        out = on_track(*a, **k)
        seen.append((hip.renderer.rgb.cpu().numpy().copy(), hip.renderer.depth.cpu().numpy().view(np.uint16).copy()))
        return out
    hip.on_track = recording
    res2 = se3.sequence.predict_sequence_ycbineoat(hip, video, str(tmp_path / "res2" / VIDEO))
    d = float(np.abs(res2["poses"] - golden["poses"]).max())
    same = sum(int(np.array_equal(a, golden["rgbA"][i]) and np.array_equal(b, golden["depthA"][i])) for i, (a, b) in enumerate(seen))
    print("  with the HIP rasteriser's image A, closed loop over %d frames: %d / %d images byte-identical to the reference renderer's, "
          "max |d pose| %.2e" % (N_FRAMES, same, len(seen), d))
    assert len(seen) == N_FRAMES and d < 1e-5
    from oracle.closed_loop import images_close as _images_close
    _images_close(seen, golden["rgbA"], golden["depthA"], exact=(0,))   # closed loop: after frame 0 the pose differs in its 7th digit
    for i in range(N_FRAMES):             # the reference run's own poses: every byte
        rgbA, depthA = hip.render_window(golden["poses_in"][i])
        assert np.array_equal(rgbA, golden["rgbA"][i]) and np.array_equal(depthA, golden["depthA"][i]), i
