"""CPU: pin the oracle (oracle/se3_oracle.py) against the goldens the REFERENCE's own code
produced (oracle/make_golden.py -> tests/golden/*.npz).  No GPU, no reference tree needed."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as Fx
from oracle import se3_oracle as O
from oracle.make_golden import LARGE_CASES, ON_TRACK_HEAD_GAIN, PRE_CASES, SUB

# oracle noise floor measured in SURVEY.md 8c: batch-1 vs batch-64 1.7e-6, layout 1.2e-6
NET_TOL = 5e-6


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_state_dict_surface():
    spec = O.state_dict_spec()
    assert len(spec) == 123
    sd = O.make_state_dict(0)
    n_f32 = sum(v.numel() for k, v in sd.items() if v.dtype == torch.float32 and "running" not in k)
    assert n_f32 == 13533446
    assert sum(1 for v in sd.values() if v.dtype == torch.int64) == 17


def test_network_matches_reference_golden(golden_dir):
    g = _load(golden_dir, "network_n3.npz")
    sd = O.make_state_dict(0)
    A, B = Fx.net_inputs(1, 3)
    # generator drift guards
    assert abs(float(A.double().sum()) - g["A_fp"][0]) < 1e-9 and float(A[0, 0, 0, 0]) == g["A_fp"][1]
    assert abs(float(sum(v.double().sum() for v in sd.values())) - g["sd_fp"][0]) < 1e-6
    out = O.forward(sd, A, B, intermediates=True)
    for k in ("trans", "rot", "trans_logit", "rot_logit"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=0, atol=NET_TOL)
    np.testing.assert_allclose(out["feature"].numpy()[:, ::SUB, ::SUB, ::SUB], g["feature"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["stemA"].numpy()[:, ::SUB, ::SUB, ::SUB], g["act_convA1"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["cat"].numpy()[:, :64][:, ::SUB, ::SUB, ::SUB], g["act_convA2"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out["trans_c2"].numpy()[:, ::SUB, ::SUB, ::SUB], g["act_trans_conv2"], rtol=1e-5, atol=2e-5)


def test_network_big_inputs(golden_dir):
    g = _load(golden_dir, "network_big_n2.npz")
    sd = O.make_state_dict(7, head_gain=0.002)
    A, B = Fx.net_inputs(11, 2, scale=40.0)
    out = O.forward(sd, A, B)
    for k in ("trans", "rot", "trans_logit", "rot_logit"):
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=0, atol=2e-5)


@pytest.mark.parametrize("case", LARGE_CASES, ids=[c[0] for c in LARGE_CASES])
def test_network_large_cases(golden_dir, case):
    """n = 8 with x40 / x150 inputs and n = 64 (BASELINE configs[1]'s batch): the reference-made logits the
    GPU tests hold the default (Winograd) path to."""
    fname, wseed, gain, iseed, n, scale = case
    g = _load(golden_dir, fname + ".npz")
    sd = O.make_state_dict(wseed, head_gain=gain)
    A, B = Fx.net_inputs(iseed, n, scale=scale)
    assert abs(float(A.double().sum()) - g["A_fp"][0]) < 1e-6 and float(A[0, 0, 0, 0]) == g["A_fp"][1]
    out = O.forward(sd, A, B)
    for k in ("trans", "rot", "trans_logit", "rot_logit"):
        assert g[k].shape == (n, 3)
        np.testing.assert_allclose(out[k].numpy(), g[k], rtol=0, atol=2e-5 if scale > 1 else NET_TOL)


@pytest.mark.parametrize("case", PRE_CASES, ids=[c[0] for c in PRE_CASES])
def test_preprocess_bit_exact(golden_dir, case):
    name, fseed, t, width = case
    g = _load(golden_dir, "preprocess.npz")
    rgb, depth = Fx.synthetic_frame(fseed)
    assert Fx.sha(rgb) + Fx.sha(depth) == str(g[name + "_frame_sha"])
    P = Fx.pose(fseed, t)
    rgbA, depthA = Fx.synthetic_render(fseed + 100, t[2])
    mean, std = Fx.mean_std(0)
    bb = O.compute_bbox(P, Fx.K_YCB, width, scale=(1000, 1000, 1000))
    assert bb.dtype == np.int32 and (bb == g[name + "_bbox"]).all()
    rgbB, depthB = O.crop_bbox(rgb, depth, bb, (176, 176))
    assert rgbB.dtype == np.uint8 and depthB.dtype == np.uint16
    assert Fx.sha(rgbB) == str(g[name + "_rgbB_sha"])
    assert Fx.sha(depthB) == str(g[name + "_depthB_sha"])
    a, b = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std)
    assert Fx.sha(a) == str(g[name + "_dataA_sha"])  # bit-exact float32
    assert Fx.sha(b) == str(g[name + "_dataB_sha"])
    assert (a[:, ::SUB, ::SUB] == g[name + "_dataA_sub"]).all()


def test_pose_update_bit_exact(golden_dir):
    g = _load(golden_dir, "pose_update.npz")
    for i in range(len(g["A"])):
        B = O.process_predict(g["A"][i], g["trans"][i], g["rot"][i], float(g["norm"][i][0]), float(g["norm"][i][1]))
        assert B.dtype == np.float64
        assert (B == g["B"][i]).all(), i
    # zero rotation -> R unchanged, bottom row exact
    assert (g["B"][0][:3, :3] == g["A"][0][:3, :3]).all()
    assert (g["B"][:, 3] == np.array([0, 0, 0, 1.0])).all()


def test_on_track_composition(golden_dir):
    g = _load(golden_dir, "on_track.npz")
    sd = O.make_state_dict(0, head_gain=ON_TRACK_HEAD_GAIN)
    mean, std = Fx.mean_std(0)
    P = Fx.pose(3)
    assert (P == g["poses"][0]).all()
    for f in range(3):
        rgb, depth = Fx.synthetic_frame(30 + f)
        rgbA, depthA = Fx.synthetic_render(130 + f, P[2, 3])
        P, info = O.on_track(sd, P, rgb, depth, rgbA, depthA, Fx.K_YCB, 250.0, mean, std, offset_rule="numpy2")   # golden made under NumPy 2
        assert (info["bbox"] == g["bbox"][f]).all()
        np.testing.assert_allclose(info["trans"], g["trans"][f], rtol=0, atol=1e-5)
        np.testing.assert_allclose(info["rot"], g["rot"][f], rtol=0, atol=1e-5)
        np.testing.assert_allclose(P, g["poses"][f + 1], rtol=0, atol=1e-6)


def test_resize_nearest_rule():
    # OpenCV resizeNN: sx = min(floor(x * (1/(dst/src))), src-1); identity when src==dst
    assert (O.resize_nearest_indices(176, 176) == np.arange(176)).all()
    idx = O.resize_nearest_indices(176, 333)
    assert idx[0] == 0 and idx[-1] == int(np.floor(175 * (1.0 / (176 / 333)))) and idx.max() <= 332
    up = O.resize_nearest_indices(176, 67)
    assert up[0] == 0 and up[-1] == 66 and (np.diff(up) >= 0).all()


def test_offset_depth_under_the_numpy_the_reference_pins(golden_dir):
    """tests/golden/preprocess_numpy1.npz: the reference's OffsetDepth / NormalizeChannels / ToTensor run UNMODIFIED under NumPy
    1.26.4 (value-based casting, like the NumPy <= 1.19 the reference pins; oracle/make_numpy1_golden.py).  The oracle's "numpy1"
    rule reproduces it bit for bit on every case, its "numpy2" rule reproduces what the same classes give under NumPy 2; the two
    differ (by <= 1 ulp of float32 before normalisation) exactly on the poses whose z * 1000 is not a float32."""
    from oracle.make_numpy1_golden import OFFSET_CASES
    g = _load(golden_dir, "preprocess_numpy1.npz")
    assert str(g["numpy_version"]).startswith("1.")
    mean, std = Fx.mean_std(0)
    cases = [(n, s, (t, w)) for n, s, t, w in PRE_CASES] + [(n, s, z) for n, s, z in OFFSET_CASES]
    ndiff_total = 0
    for name, seed, extra in cases:
        if isinstance(extra, tuple):
            t, width = extra
            rgb, depth = Fx.synthetic_frame(seed)
            P = Fx.pose(seed, t)
            rgbA, depthA = Fx.synthetic_render(seed + 100, t[2])
            rgbB, depthB = O.crop_bbox(rgb, depth, O.compute_bbox(P, Fx.K_YCB, width, scale=(1000, 1000, 1000)), (176, 176))
        else:
            P = Fx.pose(seed, (0.02, -0.01, extra))
            rgbA, depthA = Fx.synthetic_render(seed + 100, abs(extra))
            rgbB, depthB = Fx.synthetic_render(seed + 200, abs(extra))
        assert np.array_equal(P, g[name + "_pose"])
        a1, b1 = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std, offset_rule="numpy1")
        a2, b2 = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std, offset_rule="numpy2")
        assert Fx.sha(a1) == str(g[name + "_dataA_sha"]) and Fx.sha(b1) == str(g[name + "_dataB_sha"]), name
        assert Fx.sha(a2) == str(g[name + "_dataA_sha_numpy2"]) and Fx.sha(b2) == str(g[name + "_dataB_sha_numpy2"]), name
        nd = int((a1 != a2).sum() + (b1 != b2).sum())
        assert nd == int(g[name + "_dataA_ndiff_vs_numpy2"]) + int(g[name + "_dataB_ndiff_vs_numpy2"])
        assert (nd > 0) == (np.float64(np.float32(P[2, 3] * 1000)) != P[2, 3] * 1000), name
        assert np.abs(a1.astype(np.float64) - a2).max() <= 4e-6 and np.array_equal(a1[:3], a2[:3])   # depth channel only, <= 1 ulp
        ndiff_total += nd
    assert ndiff_total > 10000     # the fractional-z cases really separate the two rules
    assert O.OFFSET_RULE == "numpy1"
