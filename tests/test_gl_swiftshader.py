"""The rendered image A against a REAL OpenGL implementation (VERDICT r2 missing #1, rows a3 / f1).
tests/golden/gl_swiftshader.npz was produced by oracle/make_gl_golden.py: the UNMODIFIED reference class VispyRenderer
(vispy_renderer.py:47-178), driven as Tracker.render_window drives it (predict.py:193-208), executing on Google SwiftShader's
OpenGL ES 3.0 (Khronos-conformant software GL, shipped in the kaleido wheel of this image) through the thin vispy / PyOpenGL
stand-ins of oracle/swiftshader_gl.py.  Fill rule, clipping, perspective-correct interpolation, depth test, float -> unorm8
conversion and read-back row order are GL's own here.  Compared: the numpy restatement of the pipeline (oracle/raster_oracle.py,
CPU) and the HIP rasteriser (GPU).  Expected differences and their cause: SwiftShader snaps vertices to 1/16 pixel
(GL_SUBPIXEL_BITS = 4; the GL minimum), neither the oracle nor the HIP kernels snap -> a handful of silhouette pixels change
owner, and on meshes with pixel-sized triangles (steep colour gradients between random vertex colours) interior colours move
by 1-2 / 255."""
import os

import numpy as np
import pytest
from scipy import ndimage

from oracle import fixtures as Fx
from oracle import swiftshader_gl as SG
from oracle.make_gl_golden import CASES, OBJECT_WIDTH


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "gl_swiftshader.npz"))


def _compare(rgb, depth, want_rgb, want_d, tiny_triangles):
    assert rgb.shape == want_rgb.shape == (176, 176, 3) and depth.dtype == want_d.dtype == np.uint16
    cov, wcov = depth > 0, want_d > 0
    assert (cov != wcov).sum() <= 12, (cov != wcov).sum()                      # of 30,976 pixels: silhouette pixels, sub-pixel snapping
    both = cov & wcov
    assert both.sum() > 10000
    dd = np.abs(depth[both].astype(int) - want_d[both].astype(int))
    assert dd.max() <= 1 and (dd > 0).mean() < 0.03                            # uint16 mm truncation flips
    inner = ndimage.binary_erosion(both, iterations=2)
    d = np.abs(rgb.astype(int) - want_rgb.astype(int)).max(2)
    assert d[inner].max() <= (6 if tiny_triangles else 3), d[inner].max()
    assert np.percentile(d[inner], 99) <= 2 and (d[inner] == 0).mean() > (0.40 if tiny_triangles else 0.60)
    assert d[both & ~inner].max() <= 40                                        # rim: a different (randomly coloured) triangle owns the pixel
    assert (rgb[~cov] == 0).all() and (want_rgb[~wcov] == 0).all()            # background exactly 0 in both (maskA = depthA > 100)


def test_golden_facts(golden):
    assert "SwiftShader" in str(golden["gl_renderer"]) and str(golden["gl_version"]).startswith("OpenGL ES 3.0")
    for seed, subdiv, t in CASES:
        d = golden["depth_%d" % seed]
        z = int(round(t[2] * 1000))
        assert z - 55 <= d[d > 0].min() <= z - 40 and d.max() <= z + 15          # a 50 mm sphere: nearest point z - 50 mm, off-axis rim a little beyond z


@pytest.mark.parametrize("case", [c for c in CASES if c[1] <= 2], ids=lambda c: "seed%d" % c[0])
def test_raster_oracle_vs_real_gl(golden, case):
    from oracle import raster_oracle as R
    seed, subdiv, t = case
    m = Fx.icosphere(subdiv, 0.05, seed)
    win = tuple(int(x) for x in golden["window_%d" % seed])
    rgb, depth = R.render(m["vertices"], m["normals"].astype(np.float32), (m["colors"] / 255.0).astype(np.float32), m["faces"],
                          Fx.pose(seed, t), Fx.K_YCB, win)
    _compare(rgb, depth, golden["rgb_%d" % seed], golden["depth_%d" % seed], tiny_triangles=False)


def test_golden_is_what_swiftshader_renders_today(golden, tmp_path):
    """Where the reference tree and SwiftShader are both present (the build container): run the reference class again."""
    from oracle import ref_shims
    if not (ref_shims.reference_available() and SG.available()):
        pytest.skip("needs /root/reference and the kaleido wheel's SwiftShader")
    from oracle import make_gl_golden as M
    VR, U = M.load_reference_renderer()
    gl = SG.GL.get()
    assert gl.version == str(golden["gl_version"])
    import ctypes as C
    v = C.c_int()
    gl.glGetIntegerv(0x0D50, C.byref(v))
    assert v.value == 4                                                          # GL_SUBPIXEL_BITS (module docstring)
    for seed, subdiv, t in CASES[:3]:
        rgb, depth, win, _ = M.render_case(VR, U, str(tmp_path), seed, subdiv, t, 32)
        assert np.array_equal(rgb, golden["rgb_%d" % seed]) and np.array_equal(depth, golden["depth_%d" % seed])
        assert np.array_equal(win, golden["window_%d" % seed])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "seed%d" % c[0])
def test_hip_rasteriser_vs_real_gl(golden, case):
    import se3tracknet_amd as se3
    seed, subdiv, t = case
    eng = se3.Engine(0, 1)
    ren = se3.HipRenderer(eng, Fx.icosphere(subdiv, 0.05, seed))
    P = Fx.pose(seed, t)
    win = se3.HipRenderer.gl_window(P, Fx.K_YCB, OBJECT_WIDTH)
    assert tuple(int(x) for x in golden["window_%d" % seed]) == tuple(win)          # the reference's own compute_bbox window
    rgb, depth = ren.render(P, Fx.K_YCB, win)
    _compare(rgb, depth, golden["rgb_%d" % seed], golden["depth_%d" % seed], tiny_triangles=subdiv >= 4)


# ---- second renderer (pyrender route, full camera frame): GL's sampling / fill rules; pyrender's scene set-up is this repo's reading ------
def _compare_frame(rgb, depth, want_rgb, want_d, textured):
    assert rgb.shape == want_rgb.shape and depth.shape == want_d.shape
    cov, wcov = depth > 0, want_d > 0
    assert (cov != wcov).sum() <= 6
    both = cov & wcov
    dd = np.abs(depth[both].astype(int) - want_d[both].astype(int))
    assert dd.max() <= 2 and (dd > 1).mean() < 0.005
    inner = ndimage.binary_erosion(both, iterations=2)
    d = np.abs(rgb.astype(int) - want_rgb.astype(int)).max(2)
    if textured:   # the texture carries 5 % white speckles: single texels dominate a few pixels' level-of-detail blend
        assert np.percentile(d[inner], 99) <= 6 and np.median(d[inner]) <= 1 and (d[inner] > 8).mean() < 0.01, np.percentile(d[inner], 99)
    else:
        assert d[inner].max() <= 3
    assert (rgb[~cov] == 0).all() and (want_rgb[~wcov] == 0).all()


@pytest.mark.parametrize("i", [0, 1])
@pytest.mark.parametrize("textured", [True, False])
def test_frame_oracle_vs_real_gl(golden, i, textured):
    from oracle import raster_oracle as R
    from oracle.make_gl_golden import FRAME_HW, FRAME_K, FRAME_KD_VERTEX, FRAME_POSES
    H, W = FRAME_HW
    ms = Fx.textured_sphere(2)
    P = Fx.pose(*FRAME_POSES[i])
    v32 = ms["vertices"].astype(np.float32)
    if textured:
        rgb, depth = R.render_frame(v32, None, ms["faces"], P, FRAME_K, W, H, uv=ms["uv"], texture=ms["texture"], kd=ms["kd"])
        _compare_frame(rgb, depth, golden["frame_tex_rgb_%d" % i], golden["frame_tex_depth_%d" % i], True)
    else:
        rgb, depth = R.render_frame(v32, (ms["colors"] / 255.0).astype(np.float32), ms["faces"], P, FRAME_K, W, H, kd=FRAME_KD_VERTEX)
        _compare_frame(rgb, depth, golden["frame_vc_rgb_%d" % i], golden["frame_vc_depth_%d" % i], False)


def test_mipmap_generation_vs_real_gl(golden):
    assert golden["mipgen_max_abs_diff"][0] == 0 and golden["mipgen_max_abs_diff"].max() <= 2


@pytest.mark.gpu
@pytest.mark.parametrize("i", [0, 1])
def test_hip_full_frame_renderer_vs_real_gl(golden, i):
    import se3tracknet_amd as se3
    from oracle.make_gl_golden import FRAME_HW, FRAME_K, FRAME_KD_VERTEX, FRAME_POSES
    eng = se3.Engine(0, 1)
    ms = Fx.textured_sphere(2)
    P = Fx.pose(*FRAME_POSES[i])
    ren = se3.HipRenderer(eng, dict(vertices=ms["vertices"], faces=ms["faces"], colors=ms["colors"], uv=ms["uv"], texture=ms["texture"],
                                    kd=ms["kd"]), mode="pyrender", frame_size=FRAME_HW)
    rgb, depth = ren.render_frame(P, FRAME_K)
    _compare_frame(rgb, depth, golden["frame_tex_rgb_%d" % i], golden["frame_tex_depth_%d" % i], True)
    ren2 = se3.HipRenderer(eng, dict(vertices=ms["vertices"], faces=ms["faces"], colors=ms["colors"], kd=FRAME_KD_VERTEX), mode="pyrender",
                           frame_size=FRAME_HW)
    rgb, depth = ren2.render_frame(P, FRAME_K)
    _compare_frame(rgb, depth, golden["frame_vc_rgb_%d" % i], golden["frame_vc_depth_%d" % i], False)


@pytest.mark.gpu
def test_on_track_with_real_gl_image_A_vs_hip_image_A(golden):
    """End to end: Tracker.on_track with image A from the HIP rasteriser vs the SAME call with image A = what the reference's
    VispyRenderer rendered on real GL (injected through the renderer protocol).  The rasterisers differ in a handful of
    silhouette pixels and by 1-2 / 255 inside; this bounds what that does to the network output and the pose."""
    import se3tracknet_amd as se3
    from oracle import se3_oracle as O
    sd = O.make_state_dict(0, head_gain=0.002)
    mean, std = Fx.mean_std(0)
    worst_net = worst_pose = 0.0
    for seed, subdiv, t in CASES:
        P = Fx.pose(seed, t)
        trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH), mean, std, {"state_dict": sd})
        trk.renderer = se3.HipRenderer(trk.engine, Fx.icosphere(subdiv, 0.05, seed))
        rgb, depth = Fx.structured_frame(500 + seed)
        pose_hip = trk.on_track(P, rgb, depth)
        out_hip = np.r_[trk.last_prediction["trans"][0], trk.last_prediction["rot"][0]]

        class GoldenGL:
            def render(self, ob2cam, K, window):
                assert tuple(int(x) for x in golden["window_%d" % seed]) == tuple(window)
                return golden["rgb_%d" % seed], golden["depth_%d" % seed]
        trk.renderer = GoldenGL()
        pose_gl = trk.on_track(P, rgb, depth)
        out_gl = np.r_[trk.last_prediction["trans"][0], trk.last_prediction["rot"][0]]
        worst_net = max(worst_net, float(np.abs(out_hip - out_gl).max()))
        worst_pose = max(worst_pose, float(np.abs(pose_hip - pose_gl).max()))
    print("image A from real GL vs from the HIP rasteriser: max |d(trans, rot)| = %.2e, max |d pose| = %.2e" % (worst_net, worst_pose))
    assert worst_net < 5e-3 and worst_pose < 2e-4
