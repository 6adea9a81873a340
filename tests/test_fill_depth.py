"""fill_depth (Utils.py:455-514) as the live-camera front end applies it (predict_ros.py:38-41).
CPU: the oracle's restatement of the OpenCV calls against scipy.ndimage's independent implementations.
GPU: the HIP kernels (se3tn_fill_depth) against the oracle."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import depth_oracle as D


from oracle.fixtures import depth_frame_with_holes as _frame, depth_frame_with_far_wall as _frame_with_far_wall  # noqa: E402


def _inverted(seed):
    mm = _frame(seed)
    d = (mm / 1e3).astype(np.float32)
    v = d > 0.1
    d[v] = np.float32(2.0) - d[v]
    return d


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_morphology_and_median_vs_scipy_bitwise(seed):
    img = _inverted(seed)
    for k in (D.DIAMOND5, np.ones((5, 5), np.uint8), np.ones((7, 7), np.uint8), np.ones((31, 31), np.uint8)):
        fp = k.astype(bool)
        want = ndimage.grey_dilation(img, footprint=fp, mode="constant", cval=-np.inf)
        assert np.array_equal(D.dilate(img, k), want)
        want = ndimage.grey_erosion(img, footprint=fp, mode="constant", cval=np.inf)
        assert np.array_equal(D.erode(img, k), want)
    assert np.array_equal(D.median5(img), ndimage.median_filter(img, size=5, mode="nearest"))


def test_blurs_vs_direct_formulas():
    img = D.median5(_inverted(3))
    # Gaussian: scipy's correlate1d with 'mirror' (= BORDER_REFLECT_101) and the same fixed kernel
    k = np.array([0.0625, 0.25, 0.375, 0.25, 0.0625])
    want = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(D.gaussian5(img) - want).max() < 1e-6
    # bilateral: the table-interpolated weights against the closed form exp(-dc^2/2sc^2) exp(-r^2/2ss^2), float64
    H, W = img.shape
    p = np.pad(img.astype(np.float64), 2, mode="reflect")
    num = img.astype(np.float64).copy(); den = np.ones((H, W))
    for di in range(-2, 3):
        for dj in range(-2, 3):
            if (di == 0 and dj == 0) or di * di + dj * dj > 4:
                continue
            v = p[2 + di:2 + di + H, 2 + dj:2 + dj + W]
            w = np.exp(-0.5 * (v - img) ** 2 / 1.5 ** 2) * np.exp(-0.5 * (di * di + dj * dj) / 2.0 ** 2)
            num += v * w; den += w
    got = D.bilateral5(img, 1.5, 2.0, 5)
    assert np.abs(got - num / den).max() < 2e-6
    const = np.full((20, 30), 1.25, np.float32)
    assert np.array_equal(D.bilateral5(const), const)          # max - min < FLT_EPSILON: copied through


@pytest.mark.parametrize("extrapolate,blur", [(False, "bilateral"), (True, "bilateral"), (False, "gaussian"), (False, None)])
def test_fill_depth_properties(extrapolate, blur):
    mm = _frame(5)
    out = D.grab_depth(mm, 2.0, extrapolate, blur)
    assert out.dtype == np.uint16 and out.shape == mm.shape
    valid_in = (mm > 100) & (mm < 2000)
    # holes surrounded by surface are filled with nearby surface values; valid pixels move by at most the local relief
    inner = np.zeros_like(valid_in); inner[30:-5, 40:-5] = True
    assert (out[inner] > 100).mean() > 0.995 and (mm[inner] > 100).mean() < 0.9     # the holes are filled
    assert np.abs(out[valid_in & inner].astype(int) - mm[valid_in & inner].astype(int)).mean() < 30
    if extrapolate:
        assert (out[:, 60:] > 100).all()                 # columns with any surface are filled up to the top row
    # no value beyond the input's range is invented (max / min / median / convex blurs only)
    assert out.max() <= mm.max()


# ---- GPU ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def eng():
    import se3tracknet_amd as se3
    return se3.Engine(0, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,extrapolate", [(0, False), (1, True), (2, False)])
def test_hip_fill_depth_bit_exact_up_to_the_median(eng, seed, extrapolate):
    mm = _frame(seed, 480, 640) if seed == 2 else _frame(seed)
    got_mm, got_m = eng.fill_depth(mm, 2.0, extrapolate, blur_type=None, return_metres=True)
    want_m = D.fill_depth(mm / 1e3, 2.0, extrapolate, None)
    assert got_m.dtype == np.float32 and np.array_equal(got_m, want_m)         # selections only: bit-exact
    assert np.array_equal(got_mm, (want_m * 1000).astype(np.uint16))


def test_oracle_far_region_wraps_like_numpy_on_x86():
    mm = _frame_with_far_wall(7)
    out = D.grab_depth(mm)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = (D.fill_depth(mm / 1e3, 2.0, False, "bilateral") * 1000).astype(np.uint16)   # the reference's expression
    assert np.array_equal(out, plain)
    inner = out[110:130, 215:245]
    assert (inner > 60000).all()       # 2 m - 2.5..3 m = -0.5..-1 m -> 65536 - 500..1000: "far", invalid downstream


@pytest.mark.gpu
@pytest.mark.parametrize("blur", [None, "bilateral"])
def test_hip_fill_depth_far_region_matches_the_reference_wraparound(eng, blur):
    """a contiguous region beyond max_depth must come out as the reference's (wrapped) uint16, not clamped to 0."""
    mm = _frame_with_far_wall(7)
    got_mm, got_m = eng.fill_depth(mm, 2.0, False, blur_type=blur, return_metres=True)
    want_m = D.fill_depth(mm / 1e3, 2.0, False, blur)
    want_mm = D.grab_depth(mm, 2.0, False, blur)
    assert (got_m[110:130, 215:245] < 0).all() and np.abs(got_m - want_m).max() < 2e-6
    d = np.abs(got_mm.astype(int) - want_mm.astype(int))
    assert (got_mm[110:130, 215:245] > 60000).all()
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.gpu
def test_reserve_makes_fill_depth_and_render_frame_capturable(eng):
    """se3tn_reserve allocates the full-frame scratch at start-up: afterwards se3tn_fill_depth runs inside a hipGraph capture
    (no allocation, no synchronisation) and the replay reproduces the eager result."""
    import torch
    eng.reserve(480, 640)
    mm = _frame(2, 480, 640)
    want = eng.fill_depth(mm, 2.0, False, "bilateral")
    src = torch.from_numpy(mm.view(np.int16)).cuda()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.fill_depth(src, 2.0, False, "bilateral")       # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = eng.fill_depth(src, 2.0, False, "bilateral")
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint16), want)


@pytest.mark.gpu
@pytest.mark.parametrize("blur", ["bilateral", "gaussian"])
def test_hip_fill_depth_blurs_vs_oracle(eng, blur):
    for seed in (3, 4):
        mm = _frame(seed, 240, 320)
        got_mm, got_m = eng.fill_depth(mm, 2.0, False, blur, return_metres=True)
        want_m = D.fill_depth(mm / 1e3, 2.0, False, blur)
        err = np.abs(got_m - want_m)
        assert err.max() < 2e-6, err.max()                 # float32 sums, exp / division in the last ulp
        want_mm = (want_m * 1000).astype(np.uint16)
        d = np.abs(got_mm.astype(int) - want_mm.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3      # truncation to mm can flip on a last-ulp difference
    # device-resident input / output
    import torch
    t = torch.from_numpy(mm.view(np.int16)).cuda()
    out = eng.fill_depth(t, 2.0, False, blur)
    assert out.is_cuda and np.array_equal(out.cpu().numpy().view(np.uint16), got_mm)


def test_quaternion_from_matrix_vs_scipy():
    """predict_ros.py:63 quaternion_from_matrix (transformations, default eigen-decomposition branch), w >= 0."""
    import se3tracknet_amd as se3
    from scipy.spatial.transform import Rotation
    for i in range(300):
        r = Rotation.random(random_state=i)
        M = np.eye(4); M[:3, :3] = r.as_matrix(); M[:3, 3] = (0.1, -0.2, 0.7)
        q = se3.quaternion_from_matrix(M)
        x, y, z, w = r.as_quat()
        want = np.array([w, x, y, z]) * (1 if w >= 0 else -1)
        assert np.abs(q - want).max() < 1e-12 and q[0] >= 0


@pytest.mark.gpu
def test_live_tracker_front_end_vs_oracle():
    """TrackerRos.grab_depth / grab_color / on_track (predict_ros.py:38-66) without ROS: raw camera frames with holes
    in, (translation, quaternion xyzw) out; the pose equals the oracle's fill_depth + on_track composition."""
    import se3tracknet_amd as se3
    from oracle import fixtures as Fx
    from oracle import se3_oracle as O

    class Stub:
        def render(self, ob2cam, K, window):
            return Fx.synthetic_render(130, ob2cam[2, 3])
    sd = O.make_state_dict(0, head_gain=0.002)
    mean, std = Fx.mean_std(0)
    trk = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, renderer=Stub())
    P = Fx.pose(3)
    live = se3.LiveTracker(trk, P)
    assert live.on_track() is None                      # nothing grabbed yet
    for f in range(3):
        rgb, depth = Fx.synthetic_frame(40 + f)         # has holes (0), near and far outliers
        live.grab_depth(depth.astype(np.float64))       # any numeric dtype, as the node casts with astype(uint16)
        live.grab_color(rgb[:, :, ::-1], stamp=f * 0.033)
        want_depth = D.grab_depth(depth)
        d = np.abs(live.depth.astype(int) - want_depth.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
        trans, q_xyzw, stamp = live.on_track()
        want, aux = O.on_track(sd, P, rgb, live.depth, *Fx.synthetic_render(130, P[2, 3]), Fx.K_YCB, trk.object_width, mean, std)
        assert np.abs(live.A_in_cam - want).max() < 1e-5 and stamp == f * 0.033
        assert np.abs(np.asarray(trans) - want[:3, 3]).max() < 1e-5
        R = se3.quaternion_from_matrix(want)
        assert np.abs(np.array([R[1], R[2], R[3], R[0]]) - np.asarray(q_xyzw)).max() < 1e-5
        P = live.A_in_cam.copy()
