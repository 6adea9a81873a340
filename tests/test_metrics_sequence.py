"""Metrics (CPU, known answers from the reference's own VOCap measured in SURVEY.md section 4) and the
headless sequence driver (GPU, synthetic sequence in the reference's directory layout)."""
import os

import numpy as np
import pytest
from PIL import Image

from oracle import fixtures as Fx
from oracle import se3_oracle as O


@pytest.fixture(scope="module")
def se3():
    import se3tracknet_amd
    return se3tracknet_amd


def test_vocap_known_answers(se3):
    V = se3.metrics.VOCap
    assert V(np.zeros(10)) == pytest.approx(1.0, abs=1e-12)
    assert V(np.full(10, 0.05)) == pytest.approx(0.55, abs=1e-12)
    assert V(np.linspace(0, 0.2, 101)) == pytest.approx(0.2621782178217822, abs=1e-12)
    with pytest.raises(IndexError):
        V(np.full(5, 0.5))


def test_vocap_matches_reference_source_when_available(se3):
    from oracle import ref_shims
    if not ref_shims.reference_available():
        pytest.skip("reference tree not present")
    import importlib.util, sys, types
    ref_shims.install()
    try:
        import matplotlib.pyplot  # noqa: F401  (eval_ycb imports pyplot at module level without using it)
    except Exception:   # noqa: BLE001
        for m in ("matplotlib", "matplotlib.pyplot"):
            sys.modules.setdefault(m, types.ModuleType(m))
    spec = importlib.util.spec_from_file_location("ref_eval_ycb", os.path.join(ref_shims.REFERENCE_ROOT, "eval_ycb.py"))
    try:
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    except Exception as e:  # noqa
        pytest.skip("reference eval_ycb not importable here: %r" % (e,))
    rng = np.random.default_rng(0)
    for _ in range(20):
        errs = rng.exponential(0.03, 200)
        # own vectorised implementation vs the reference's loop: equal up to float64 summation order
        assert se3.metrics.VOCap(errs) == pytest.approx(mod.VOCap(errs), abs=1e-13)
        tied = np.round(errs, 2)                  # many exact ties, some exactly 0 and exactly 0.1
        assert se3.metrics.VOCap(tied) == pytest.approx(mod.VOCap(tied), abs=1e-13)
    assert se3.metrics.auc(np.full(5, 0.5)) == 0.0 and se3.metrics.auc([]) == 0.0


def test_add_adi_vs_brute_force(se3):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-0.05, 0.05, (400, 3))
    P, G = Fx.pose(1), Fx.pose(2)
    a = pts @ P[:3, :3].T + P[:3, 3]; b = pts @ G[:3, :3].T + G[:3, 3]
    assert se3.metrics.add(P, G, pts) == pytest.approx(np.linalg.norm(a - b, axis=1).mean(), abs=1e-15)
    brute = np.sqrt(((b[:, None, :] - a[None, :, :]) ** 2).sum(-1)).min(1).mean()
    assert se3.metrics.adi(P, G, se3.utils.PointCloud(pts)) == pytest.approx(brute, abs=1e-12)
    assert se3.metrics.adi(P, P, pts) == 0.0 and se3.metrics.add(P, P, pts) == 0.0


class _Render:
    def render(self, ob2cam, K, window):
        return Fx.synthetic_render(11, ob2cam[2, 3])


@pytest.mark.gpu
def test_sequence_driver_on_synthetic_ycb_layout(se3, tmp_path):
    seq = tmp_path / "0048"
    for d in ("color", "depth_filled", "pose_gt/4"):
        os.makedirs(seq / d)
    P = Fx.pose(3)
    n = 6
    for i in range(n):
        rgb, depth = Fx.synthetic_frame(60 + i)
        Image.fromarray(rgb).save(seq / "color" / ("%06d.png" % i))
        Image.fromarray(depth).save(seq / "depth_filled" / ("%06d.png" % i))
        np.savetxt(seq / "pose_gt/4" / ("%06d.txt" % i), P)
    sd = O.make_state_dict(0, head_gain=0.0005)
    mean, std = Fx.mean_std(0)
    ply = tmp_path / "m.ply"
    pts = np.random.default_rng(0).uniform(-0.04, 0.04, (300, 3))
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 300\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        f.writelines("%.8f %.8f %.8f\n" % tuple(p) for p in pts)
    info = dict(Fx.DATASET_INFO); info.pop("object_width")
    trk = se3.Tracker(info, mean, std, {"state_dict": sd}, model_path=str(ply), renderer=_Render())
    assert 80 < trk.object_width < 200  # 1.1 x hull diameter of an 8 cm cube cloud, in mm
    out = tmp_path / "out"
    res = se3.sequence.predict_sequence_ycb(trk, str(seq), 4, str(out) + "/")
    assert res["frames"] == n - 1 and res["poses"].shape == (n, 4, 4) and res["hz"] > 50
    assert sorted(os.listdir(out))[:2] == ["00000.txt", "00000gt.txt"] and len(os.listdir(out)) == 2 * n
    assert np.allclose(np.loadtxt(out / "00000.txt"), P)
    assert np.allclose(np.loadtxt(out / ("%05d.txt" % (n - 1))), res["poses"][-1])
    # replay frame 1 through the oracle: same pose
    rgb1, depth1 = Fx.synthetic_frame(61)
    want, _ = O.on_track(sd, P, rgb1, depth1, *Fx.synthetic_render(11, P[2, 3]), Fx.K_YCB, trk.object_width, mean, std)
    assert np.abs(res["poses"][1] - want).max() < 1e-5
    assert 0.0 <= res["adi_auc"] <= 100.0 and len(res["adi_errs"]) == n


@pytest.mark.gpu
def test_get_results_and_eval_one_class_file_formats(se3, tmp_path):
    """getResultsYcb layout in, eval_ycb layout out: seq<ID>/%07d.txt with file index = frame id - 1,
    keyframe filter, CADmodels/*/points.xyz -- on a synthetic data_organized tree."""
    ycb = tmp_path / "ycb"
    P = Fx.pose(3)
    for seq, nfr in ((48, 4), (50, 3), (10, 2)):      # 0010 is a training video: skipped (predict.py:349)
        for d in ("color", "depth_filled", "pose_gt/2"):
            os.makedirs(ycb / "data_organized" / ("%04d" % seq) / d)
        for i in range(nfr):
            rgb, depth = Fx.synthetic_frame(100 + seq + i)
            Image.fromarray(rgb).save(ycb / "data_organized" / ("%04d" % seq) / "color" / ("%06d.png" % (i + 1)))
            Image.fromarray(depth).save(ycb / "data_organized" / ("%04d" % seq) / "depth_filled" / ("%06d.png" % (i + 1)))
            np.savetxt(ycb / "data_organized" / ("%04d" % seq) / "pose_gt/2" / ("%06d.txt" % (i + 1)), P)
    os.makedirs(ycb / "data_organized" / "0049" / "color")            # a sequence without this class
    for c in ("001_a", "002_b"):
        os.makedirs(ycb / "CADmodels" / c)
        np.savetxt(ycb / "CADmodels" / c / "points.xyz", np.random.default_rng(1).uniform(-0.04, 0.04, (200, 3)))
    os.makedirs(ycb / "YCB_Video_toolbox")
    (ycb / "YCB_Video_toolbox" / "keyframe.txt").write_text("0048/000001\n0048/000003\n0050/000002\n0051/000001\n")
    sd = O.make_state_dict(0, head_gain=0.0005)
    mean, std = Fx.mean_std(0)
    trk = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, renderer=_Render())
    out = str(tmp_path / "res") + "/"
    done = se3.sequence.get_results_ycb(trk, str(ycb), 2, out)
    assert done == {48: 4, 50: 3}
    assert sorted(os.listdir(out)) == ["seq48", "seq50"] and sorted(os.listdir(out + "seq48"))[0] == "0000000.txt"
    assert np.allclose(np.loadtxt(out + "seq48/0000000.txt"), P)     # GT initialisation
    res = se3.sequence.eval_one_class(out, str(ycb), 2)
    assert res["n"] == 3 and 0 <= res["adi_auc"] <= 100 and res["adi_errs"][0] == 0.0   # frame 1 == GT


@pytest.mark.gpu
def test_ycbineoat_driver_and_eval_file_formats(se3, tmp_path):
    """predictSequenceYcbInEOAT / eval_ycbineoat.py layouts on a synthetic tree: rgb/ depth_filled/
    annotated_poses/ in, %07d.txt out (frame 0 tracked too, rot_normalizer 30 deg), results folder
    matched to its object by name."""
    data = tmp_path / "eoat"
    vid = "mustard0"
    for d in ("rgb", "depth_filled", "annotated_poses"):
        os.makedirs(data / vid / d)
    P = Fx.pose(3)
    n = 4
    for i in range(n):
        rgb, depth = Fx.synthetic_frame(200 + i)
        Image.fromarray(rgb).save(data / vid / "rgb" / ("%07d.png" % i))
        Image.fromarray(depth).save(data / vid / "depth_filled" / ("%07d.png" % i))
        np.savetxt(data / vid / "annotated_poses" / ("%07d.txt" % i), P)
    ycb = tmp_path / "ycb"
    for c in ("003_cracker_box", "006_mustard_bottle"):
        os.makedirs(ycb / "CADmodels" / c)
        np.savetxt(ycb / "CADmodels" / c / "points.xyz", np.random.default_rng(2).uniform(-0.04, 0.04, (150, 3)))
    sd = O.make_state_dict(0, head_gain=0.0005)
    mean, std = Fx.mean_std(0)
    trk = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, renderer=_Render(),
                      trans_normalizer=0.03, rot_normalizer=30 * np.pi / 180)
    res_dir = tmp_path / "res"
    out = se3.sequence.predict_sequence_ycbineoat(trk, str(data / vid), str(res_dir / vid))
    assert out["poses"].shape == (n, 4, 4) and sorted(os.listdir(res_dir / vid)) == ["%07d.txt" % i for i in range(n)]
    # frame 0 is a tracked frame (not the GT copy), with the 30-degree rotation normaliser
    rgb0, depth0 = Fx.synthetic_frame(200)
    want, _ = O.on_track(sd, P, rgb0, depth0, *Fx.synthetic_render(11, P[2, 3]), Fx.K_YCB, trk.object_width, mean, std,
                         rot_normalizer=30 * np.pi / 180)
    assert np.abs(out["poses"][0] - want).max() < 1e-5 and np.abs(out["poses"][0] - P).max() > 0
    ev = se3.sequence.eval_ycbineoat(str(res_dir), str(data), str(ycb))
    assert list(ev["per_object"]) == ["mustard"] and ev["n"] == n and 0 <= ev["adi_auc"] <= 100
