"""Third-party ground truth, the moment it exists.  oracle/pin_opencv.py / oracle/pin_gl.py are run ONCE on a machine that has
opencv-python / the reference's GL stack and write tests/golden/opencv_*.npz / gl_*.npz; until then these tests skip (and the
rows stay "parity unpinned").  When the files are present, the numpy oracle AND the HIP kernels are
compared with the real cv2 / GL outputs.  What runs here and now: the pin script's statement of the fill_depth chain, driven
by a stand-in `cv2` built from the oracle's operators, reproduces the oracle -- so script and oracle describe the same chain."""
import importlib.util
import os
import types

import numpy as np
import pytest

from oracle import depth_oracle as D
from oracle import fixtures as Fx
from oracle import se3_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _golden(golden_dir, name):
    p = os.path.join(golden_dir, name)
    if not os.path.isfile(p):
        pytest.skip("%s not pinned yet: run oracle/pin_opencv.py / oracle/pin_gl.py where the third-party stack exists" % name)
    return np.load(p, allow_pickle=False)


def test_pin_script_chain_equals_the_oracle_chain():
    pin = _load("pin_opencv")
    fake = types.SimpleNamespace(MORPH_CLOSE=3, dilate=D.dilate,
                                 morphologyEx=lambda img, op, k: D.erode(D.dilate(img, k), k),
                                 medianBlur=lambda img, k: D.median5(img), GaussianBlur=lambda img, ks, s: D.gaussian5(img),
                                 bilateralFilter=lambda img, d, sc, ss: D.bilateral5(img, sc, ss, d))
    for seed, H, W, extrap, blur in pin.FILL_CASES:
        mm = Fx.depth_frame_with_far_wall(7) if seed == "wall7" else Fx.depth_frame_with_holes(seed, H, W)
        if H * W > 100000:
            mm = mm[:200, :240]
        st = pin.fill_depth_stages(fake, mm / 1e3, 2.0, extrap, blur)
        stages = {}
        want = D.fill_depth(mm / 1e3, 2.0, extrap, blur, stages=stages)
        assert np.array_equal(st["out_m"], want), (seed, blur)
        assert np.array_equal(st["median5"], stages["median"])
        assert np.array_equal(st["fill31" if extrap else "fill7"], stages["filled"])
        assert np.array_equal((st["out_mm"].astype(np.int64)) & 0xFFFF, D.grab_depth(mm, 2.0, extrap, blur).astype(np.int64))


def test_pin_gl_helpers_match_the_oracle():
    gl = _load("pin_gl")
    P = Fx.pose(1, (-0.05, 0.04, 0.9))
    assert np.array_equal(gl.compute_bbox(P, Fx.K_YCB, 130.0, (1000, -1000, 1000)), O.compute_bbox(P, Fx.K_YCB, 130.0, (1000, -1000, 1000)))


# ---- active once the goldens exist ---------------------------------------------------------------------------------------
def test_resize_nearest_rule_vs_real_opencv(golden_dir):
    g = _golden(golden_dir, "opencv_resize.npz")
    for src in range(1, 2001):
        want = O.resize_nearest_indices(176, src)
        assert np.array_equal(want, g["index_x"][src - 1]) and np.array_equal(want, g["index_y"][src - 1]), src


def test_rodrigues_vs_real_opencv(golden_dir):
    g = _golden(golden_dir, "opencv_rodrigues.npz")
    for v, R in zip(g["rvec"], g["R"]):
        got = O.rodrigues(v)
        assert got.dtype == R.dtype and np.array_equal(got, R), v


def test_fill_depth_oracle_vs_real_opencv(golden_dir):
    g = _golden(golden_dir, "opencv_fill_depth.npz")
    pin = _load("pin_opencv")
    for seed, H, W, extrap, blur in pin.FILL_CASES:
        mm = Fx.depth_frame_with_far_wall(7) if seed == "wall7" else Fx.depth_frame_with_holes(seed, H, W)
        stages = {}
        out = D.fill_depth(mm / 1e3, 2.0, extrap, blur, stages=stages)
        assert np.array_equal(stages["median"], g["%s_median5" % seed]), seed        # selections: bit-exact
        assert np.abs(out - g["%s_out_m" % seed]).max() < 2e-6, seed                 # blurs: float32 summation order
        assert np.array_equal(D.grab_depth(mm, 2.0, extrap, blur), g["%s_out_mm" % seed]) or \
            np.abs(D.grab_depth(mm, 2.0, extrap, blur).astype(int) - g["%s_out_mm" % seed].astype(int)).max() <= 1


@pytest.mark.gpu
def test_hip_fill_depth_vs_real_opencv(golden_dir):
    g = _golden(golden_dir, "opencv_fill_depth.npz")
    import se3tracknet_amd as se3
    pin = _load("pin_opencv")
    eng = se3.Engine(0, 1)
    for seed, H, W, extrap, blur in pin.FILL_CASES:
        mm = Fx.depth_frame_with_far_wall(7) if seed == "wall7" else Fx.depth_frame_with_holes(seed, H, W)
        got_mm, got_m = eng.fill_depth(mm, 2.0, extrap, blur_type=blur, return_metres=True)
        assert np.abs(got_m - g["%s_out_m" % seed]).max() < 4e-6, seed
        d = np.abs(got_mm.astype(int) - g["%s_out_mm" % seed].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3


def _vispy_cases():
    return _load("pin_gl").VISPY_CASES


def test_raster_oracle_vs_real_vispy(golden_dir):
    g = _golden(golden_dir, "gl_vispy.npz")
    from oracle import raster_oracle as R
    for seed, subdiv, t, width in _vispy_cases():
        m = Fx.icosphere(subdiv, 0.05, seed)
        P = Fx.pose(seed, t)
        rgb, depth = R.render(m["vertices"], m["normals"].astype(np.float32), (m["colors"] / 255.0).astype(np.float32), m["faces"],
                              P, Fx.K_YCB, tuple(int(x) for x in g["window_%d" % seed]))
        want_rgb, want_d = g["rgb_%d" % seed], g["depth_%d" % seed]
        assert ((depth > 0) == (want_d > 0)).mean() > 0.998
        both = (depth > 0) & (want_d > 0)
        assert np.abs(depth[both].astype(int) - want_d[both].astype(int)).max() <= 1
        assert np.abs(rgb[both].astype(int) - want_rgb[both].astype(int)).max() <= 2


@pytest.mark.gpu
def test_hip_rasteriser_vs_real_vispy(golden_dir):
    g = _golden(golden_dir, "gl_vispy.npz")
    import se3tracknet_amd as se3
    eng = se3.Engine(0, 1)
    for seed, subdiv, t, width in _vispy_cases():
        ren = se3.HipRenderer(eng, Fx.icosphere(subdiv, 0.05, seed))
        rgb, depth = ren.render(Fx.pose(seed, t), Fx.K_YCB, tuple(int(x) for x in g["window_%d" % seed]))
        want_rgb, want_d = g["rgb_%d" % seed], g["depth_%d" % seed]
        assert ((depth > 0) == (want_d > 0)).mean() > 0.998
        both = (depth > 0) & (want_d > 0)
        assert np.abs(depth[both].astype(int) - want_d[both].astype(int)).max() <= 1
        assert np.abs(rgb[both].astype(int) - want_rgb[both].astype(int)).max() <= 2


@pytest.mark.gpu
def test_hip_full_frame_renderer_vs_real_pyrender(golden_dir):
    g = _golden(golden_dir, "gl_pyrender.npz")
    import se3tracknet_amd as se3
    gl = _load("pin_gl")
    eng = se3.Engine(0, 1)
    m = Fx.icosphere(2, 0.05, 3)
    ren = se3.HipRenderer(eng, dict(vertices=m["vertices"], faces=m["faces"], colors=m["colors"]), mode="pyrender",
                          frame_size=gl.PYR_HW)
    for i, t in enumerate([(0.01, -0.02, 0.45), (-0.03, 0.02, 0.7)]):
        rgb, depth = ren.render_frame(Fx.pose(4 + i, t), gl.PYR_K)
        want_rgb, want_d = g["rgb_%d" % i], g["depth_%d" % i]
        assert ((depth > 0) == (want_d > 0)).mean() > 0.998
        both = (depth > 0) & (want_d > 0)
        assert np.abs(depth[both].astype(int) - want_d[both].astype(int)).max() <= 1
        assert np.median(np.abs(rgb[both].astype(int) - want_rgb[both].astype(int))) <= 1
