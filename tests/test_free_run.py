"""Free-running two-track comparison (oracle/free_run.py): the HIP tracker's own closed loop beside the
oracle's own closed loop, predict.py:416-420 feedback on both sides, nothing shared but the start pose and the camera frames.
CPU part: the comparison machinery against the oracle itself (identical runs -> zero separation; the channels-last control ->
a separation that starts at rounding level).  GPU part: the bounds the measured figures support."""
import numpy as np
import pytest

from oracle import closed_loop as CL, free_run as FR


@pytest.fixture(scope="module")
def cpu_tracks():
    from oracle import raster_oracle as R
    mesh = R.icosphere(3, 0.06, 0)
    pb = FR.RandomInitProblem(1, "ycbineoat_30deg", mesh=mesh)
    job = dict(problem=pb.spec(), frames=5, threads=4)
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "a")
        a = FR.oracle_track(dict(job, save_to=path))
        b = FR.oracle_track(dict(job, images=path))
        c = FR.oracle_track(dict(job, images=path, variant="channels_last"))
    return mesh, a, b, c


def test_identical_oracle_runs_do_not_separate(cpu_tracks):
    mesh, a, b, _ = cpu_tracks
    r = FR.compare_tracks(a, b, np.asarray(mesh["vertices"]), None)
    assert r["frames"] == 5 and r["first_bbox_divergence_frame"] is None and r["bbox_differing_frames"] == 0
    assert r["max_abs_pose_separation"] == 0.0 and r["first_pose_divergence_frame"] is None
    assert r["imageA_differing_frames"] == 0 and r["first_imageA_divergence_frame"] is None
    assert r["frames_within_1e-5"] == 5 and r["reinits"] == [0, 0]


def test_control_track_starts_at_rounding_level_and_reports_every_field(cpu_tracks):
    import importlib
    metrics = importlib.import_module("iros20-6d-pose-tracking_amd.metrics") if _has_lib() else None
    mesh, a, _, c = cpu_tracks
    r = FR.compare_tracks(a, c, np.asarray(mesh["vertices"])[::4], metrics)
    # frame 0 is fed the same pose on both sides: its outputs differ by the summation order only
    assert np.array_equal(a["poses"][0], c["poses"][0]) and np.array_equal(a["bboxes"][0], c["bboxes"][0])
    d0 = float(np.abs(a["outs"][0] - c["outs"][0]).max())
    assert 0.0 < d0 < 1e-4, d0
    assert r["pose_separation_at_frame"]["1"] < 1e-5 and r["first_pose_divergence_frame"] == 1
    assert len(r["pose_separation_by_window_of_100"]) == 1 and len(r["imageA_differing_pixels_by_window_of_100"]) == 1
    if metrics is not None:
        assert r["adds_between_tracks_mm"]["max"] <= r["add_between_tracks_mm"]["max"] + 1e-12
        assert 0.0 <= r["adds_auc_vs_oracle_track"] <= 100.0
    s = FR.summarise({"seed_1": r})
    assert s["tracks"] == 1 and s["frames_total"] == 5 and s["max_abs_pose_separation"] == r["max_abs_pose_separation"]


def _has_lib():
    try:
        import se3tracknet_amd  # noqa: F401
        return True
    except (ImportError, OSError):
        return False


def test_per_seed_setups_differ_and_feedback_rule_matches_closed_loop():
    assert len({tuple(FR.initial_pose(s).reshape(-1)) for s in range(3)}) == 3
    P = FR.initial_pose(0)
    Q = P.copy(); Q[:3, 3] += (0.001, -0.002, 0.003)
    n0, r0 = FR.next_pose(P, Q, 7, 0)
    n1, r1 = CL._next_pose(P, Q, 7)
    assert np.array_equal(n0, n1) and r0 == r1 == 0
    Q[2, 3] = 5.0                                        # out of the frustum: re-detected at the anchor
    n2, r2 = FR.next_pose(P, Q, 7, 37)
    assert r2 == 1 and np.allclose(n2[:3, 3], CL.anchor(8 + 37))


def test_rotation_angle_small_and_large():
    from scipy.spatial.transform import Rotation
    Ra = Rotation.from_rotvec([[0.1, 0.2, 0.3]] * 3).as_matrix()
    Rb = np.stack([Rotation.from_rotvec(v).as_matrix() for v in ([1e-9, 0, 0], [0, 0.5, 0], [0, 0, 3.0])]) @ Ra
    ang = FR._rot_angle_deg(Rb, Ra)
    assert np.allclose(ang, np.degrees([1e-9, 0.5, 3.0]), rtol=1e-6)


# ------------------------------------------------------------------------------------------------------------------------------
# GPU: the HIP tracker's own closed loop beside the oracle's own closed loop
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def report():
    import os
    import se3tracknet_amd as se3
    if not os.path.exists(FR.default_synth_weights()):
        pytest.skip("tests/golden/synth_tracker.npz not generated")
    return FR.run_report(se3, frames_tracked=300, frames_random=300, seeds=(0, 1), control_seeds=(0, 1))


@pytest.mark.gpu
def test_free_running_tracks_with_trained_weights_stay_together(report):
    """The track-level claim, measured at 1000 frames x 3 seeds in profiles/r06_free_run.json and bounded here at
    300 x 2: with weights that TRACK (a contractive loop, as with the reference's pretrained weights), two independent runs of
    predict.py:416-420 -- HIP tracker / CPU oracle -- from the same start over the same frames
      * do NOT stay bit-identical, and neither does the reference against ITSELF: the first differing integer crop window comes after
        20-50 frames in both pairs (a rounding difference flips one np.round in compute_bbox; the crop is then resampled one pixel over),
      * stay TOGETHER: the separation is bounded for the whole sequence (no accumulation: the maximum of every 100-frame window is of
        one size), at the level at which the oracle separates from itself under a change of memory format (median 7e-4, max 0.02 on the
        4x4; 1 mm / 1 degree), far below the tracker's own error against the ground truth,
      * score the same: ADD / ADD-S AUC against the ground truth equal to 0.01 (eval_ycb.py:45-119), as the oracle vs itself."""
    _assert_tracked(report["synthetic_tracking_trained_weights"], "ycbineoat_30deg")


@pytest.mark.gpu
def test_free_running_tracks_with_trained_weights_5_degree_regime(report):
    """the same under the YCB-Video normalisers of predict.py:128 (0.03 m, 5 degrees; its own trained stand-in, 1-1.5 degrees per frame)"""
    if "synthetic_tracking_trained_weights_5deg" not in report:
        pytest.skip("tests/golden/synth_tracker_5deg.npz not generated")
    _assert_tracked(report["synthetic_tracking_trained_weights_5deg"], "ycb_video_5deg")


def _assert_tracked(r, regime):
    assert r["regime"] == regime
    ctl, hc = r["control_oracle_vs_oracle_channels_last"], r["hip_vs_oracle_channels_last"]
    print({k: v for k, v in r.items() if k not in ("tracks_detail", "control_oracle_vs_oracle_channels_last", "hip_vs_oracle_channels_last", "what")})
    print("control", {k: v for k, v in ctl.items() if k != "tracks_detail"})
    assert r["tracks"] == 2 and r["frames_per_track"] == 300 and r["reinits"] == [0, 0] and ctl["reinits"] == [0, 0]
    # bounded, and of the control's size (the control is itself a sample of a noisy quantity: factor 3)
    assert r["max_abs_pose_separation"] <= 0.05 and r["max_abs_pose_separation"] <= 3 * ctl["max_abs_pose_separation"], (r["max_abs_pose_separation"], ctl["max_abs_pose_separation"])
    assert r["median_abs_pose_separation"] <= 3 * ctl["median_abs_pose_separation"] + 1e-4
    assert r["max_rotation_separation_deg"] <= 3.0 and r["max_translation_separation_mm"] <= 3.0
    assert r["bbox_differing_frames"] <= 1.5 * ctl["bbox_differing_frames"] + 30
    assert r["earliest_bbox_divergence_frame"] is None or r["earliest_bbox_divergence_frame"] >= 5
    for seed, t in r["tracks_detail"].items():
        w = t["pose_separation_by_window_of_100"]
        assert max(w) <= 6 * float(np.median(w)) + 1e-4, (seed, w)                          # no growth over the sequence
        assert t["pose_separation_at_frame"]["1"] <= 1e-5                                    # the first step is at rounding level
        assert t["add_between_tracks_mm"]["max"] < 3.0 and t["adds_between_tracks_mm"]["max"] < 3.0, (seed, t)
        assert t["adds_auc_vs_oracle_track"] > 99.5
    for seed, g in r["against_ground_truth"].items():
        # the tracker tracks (AUC of ADD-S against the ground truth: eval_ycb.py:45-119) and both implementations score the same
        assert g["hip"]["reinits"] == 0 and g["oracle"]["reinits"] == 0
        # (measured: AUC 99.1-99.2 under 30 degrees, 99.2-99.3 under 5)
        assert g["oracle"]["adds_auc"] > 98.0 and g["hip"]["adds_auc"] > 98.0 and g["hip"]["add_auc"] > 98.0, g
        assert g["hip"]["adds_mm_max"] < 5.0 and g["oracle"]["adds_mm_max"] < 5.0
        assert abs(g["adds_auc_hip_minus_oracle"]) < 0.05 and abs(g["add_auc_hip_minus_oracle"]) < 0.05, g
    assert r["hz_hip"] > 1000


@pytest.mark.gpu
def test_free_running_random_init_diverges_no_faster_than_the_reference_from_itself(report):
    """Random-init weights have no restoring force: the loop pose -> image A -> network -> pose amplifies a rounding difference until
    the integer crop window flips (5-20 frames), after which the tracks are unrelated -- for the oracle against ITSELF (channels-last
    inputs) exactly as for the HIP tracker against the oracle.  What is asserted: the start is at per-step rounding level, and the HIP
    pair does not separate earlier than the control pair does (same order of frames)."""
    r = report["random_init"]
    for regime, blk in r["regimes"].items():
        ctl = blk["control_oracle_vs_oracle_channels_last"]
        first = [t["first_bbox_divergence_frame"] for t in blk["tracks_detail"].values()]
        first_c = [t["first_bbox_divergence_frame"] for t in ctl["tracks_detail"].values()]
        print(regime, "first bbox divergence HIP-vs-oracle", first, "oracle-vs-oracle", first_c)
        assert blk["reinits"] == [0, 0] and ctl["reinits"] == [0, 0]
        for t in blk["tracks_detail"].values():
            assert t["pose_separation_at_frame"]["1"] <= 1e-5 and t["frames_within_1e-5"] >= 2, t["pose_separation_at_frame"]
        assert all(f is not None and f >= 3 for f in first), first
        assert min(first) >= 0.33 * min(first_c), (first, first_c)
