"""Rasteriser: the HIP kernels against the CPU statement of the GL implementation the goldens were
rendered on (oracle/ss_rules.py, confirmed bit for bit against the live library: tests/test_ss_rules.py) -- byte equality --
plus properties and an analytic sphere.  oracle/raster_oracle.py is the older float-coordinate restatement of the GL pipeline
(no sub-pixel snapping); it stays as an independent cross-check of the geometry."""
import numpy as np
import pytest
import torch

from oracle import fixtures as Fx
from oracle import raster_oracle as R
from oracle import ss_rules as S


@pytest.fixture(scope="module")
def se3():
    import se3tracknet_amd
    return se3tracknet_amd


def _mesh_args(m):
    nrm = m["normals"] / np.linalg.norm(m["normals"], axis=1, keepdims=True)
    return m["vertices"].astype(np.float32), nrm.astype(np.float32), (m["colors"] / 255.0).astype(np.float32), m["faces"]


def test_oracle_projection_geometry():
    """The GL matrices of update_cam_mat map camera point (x,y,z) to window x = (u - left) W/(right-left),
    and the read-back row order is top-down in the OpenCV image."""
    K = Fx.K_YCB
    m = R.icosphere(1, 0.04)
    P = np.eye(4); P[:3, 3] = (0.02, -0.03, 0.7)
    win = (250, 190, 420, 360)  # left, top, right, bottom in (u, cy - fy y/z)
    rgb, depth = R.render(*_mesh_args(m), P, K, win)
    ys, xs = np.nonzero(depth)
    u = K[0, 0] * 0.02 / 0.7 + K[0, 2]; vflip = K[1, 2] - K[1, 1] * (-0.03) / 0.7
    cx_px = (u - win[0]) * 176 / (win[2] - win[0])
    row_px = (win[3] - vflip) * 176 / (win[3] - win[1])  # rows count down from `bottom`
    assert abs(xs.mean() + 0.5 - cx_px) < 1.0 and abs(ys.mean() + 0.5 - row_px) < 1.0
    assert 700 - 41 <= depth[depth > 0].min() and depth.max() <= 700 + 1
    # object above the optical axis in the OpenCV image (y < 0 => small v) appears in the upper half
    P2 = np.eye(4); P2[:3, 3] = (0.0, -0.05, 0.7)
    full = (0, int(2 * K[1, 2]) - 480, 640, int(2 * K[1, 2]))
    _, d2 = R.render(*_mesh_args(m), P2, K, full)
    assert np.nonzero(d2)[0].mean() < 88


@pytest.mark.gpu
# (3, 0, ...): 20 large triangles, each a bounding box of thousands of pixels: the wave-per-triangle path (raster_big_kernel)
@pytest.mark.parametrize("seed,subdiv,t", [(0, 2, (0.03, -0.02, 0.65)), (1, 3, (-0.05, 0.04, 0.9)), (2, 1, (0.0, 0.0, 0.45)),
                                           (3, 0, (0.005, -0.004, 0.3))])
def test_hip_rasteriser_vs_oracle(se3, seed, subdiv, t):
    m = R.icosphere(subdiv, 0.05, seed)
    eng = se3.Engine(0, 1)
    ren = se3.HipRenderer(eng, m)
    P = Fx.pose(seed, t)
    win = se3.HipRenderer.gl_window(P, Fx.K_YCB, 130.0)
    rgb, depth = ren.render(P, Fx.K_YCB, win)
    orgb, odepth = S.render_vispy(*_mesh_args(m), P, Fx.K_YCB, win, numpy_rule="numpy1")
    assert rgb.shape == (176, 176, 3) and depth.dtype == np.uint16
    assert (depth > 0).sum() > 2000
    assert np.array_equal(depth, odepth) and np.array_equal(rgb, orgb)       # every byte
    assert (rgb[~(depth > 0)] == 0).all()                  # background exactly 0 (Tracker's maskA = depthA > 100)
    # the float-coordinate restatement of the pipeline agrees up to what 1/16-pixel snapping moves
    frgb, fdepth = R.render(*_mesh_args(m), P, Fx.K_YCB, win)
    assert ((depth > 0) == (fdepth > 0)).mean() > 0.9995
    both = (depth > 0) & (fdepth > 0)
    assert np.abs(depth[both].astype(int) - fdepth[both].astype(int)).max() <= 1
    # deterministic (atomicMin on (depth | triangle id) keys)
    rgb2, depth2 = ren.render(P, Fx.K_YCB, win)
    assert (rgb2 == rgb).all() and (depth2 == depth).all()


@pytest.mark.gpu
def test_tracker_with_builtin_renderer(se3, tmp_path):
    """Tracker without an injected renderer: model .ply with faces -> HipRenderer; rendered A stays on
    the device.  Checks on_track against the oracle composition fed with the same render."""
    from oracle import se3_oracle as O
    m = R.icosphere(3, 0.06, 5)
    ply = tmp_path / "obj.ply"
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "property float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\n"
                "property uchar blue\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(m["vertices"]), len(m["faces"])))
        for v, n, c in zip(m["vertices"], m["normals"], m["colors"]):
            f.write("%.8f %.8f %.8f %.8f %.8f %.8f %d %d %d\n" % (*v, *n, *c))
        for a, b, c in m["faces"]:
            f.write("3 %d %d %d\n" % (a, b, c))
    sd = O.make_state_dict(0, head_gain=0.002)
    mean, std = Fx.mean_std(0)
    info = dict(Fx.DATASET_INFO); info["object_width"] = 150.0
    trk = se3.Tracker(info, mean, std, {"state_dict": sd}, model_path=str(ply))
    assert isinstance(trk.renderer, se3.HipRenderer)
    rgb, depth = Fx.synthetic_frame(12)
    P0 = Fx.pose(3, (0.02, 0.01, 0.7))
    rgbA, depthA = trk.render_window(P0)
    assert rgbA.shape == (176, 176, 3) and (depthA > 100).sum() > 3000
    P1 = trk.on_track(P0, rgb, depth)
    want, _ = O.on_track(sd, P0, rgb, depth, rgbA, depthA, Fx.K_YCB, 150.0, mean, std)
    assert np.abs(P1 - want).max() < 1e-5


# ---- the reference's second renderer (pyrender, textured .obj): offscreen_renderer.py:48-83, predict.py:161-164,209-213 ----
def _textured_sphere(tmp_path, subdiv=2, radius=0.05, tex_hw=(64, 128)):
    """icosphere with spherical texture coordinates, written as .obj + .mtl + .png; returns (paths, mesh dict)."""
    from PIL import Image
    ms = Fx.textured_sphere(subdiv, radius, tex_hw)
    m, v, uv, tex = ms, ms["vertices"], ms["uv"], ms["texture"]
    Image.fromarray(tex).save(tmp_path / "tex.png")
    (tmp_path / "obj.mtl").write_text("newmtl m0\nKa 0.2 0.2 0.2\nKd 0.9 1.0 0.8\nmap_Kd tex.png\n")
    with open(tmp_path / "obj.obj", "w") as f:
        f.write("mtllib obj.mtl\nusemtl m0\n")
        for p in v:
            f.write("v %.9f %.9f %.9f\n" % tuple(p))
        for t in uv:
            f.write("vt %.9f %.9f\n" % tuple(t))
        for a, b, c in m["faces"]:
            f.write("f %d/%d %d/%d %d/%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1))
    return str(tmp_path / "obj.obj"), dict(vertices=v, faces=m["faces"], uv=uv, texture=tex, kd=np.array([0.9, 1.0, 0.8]))


def test_obj_loader_and_mip_pyramid(se3, tmp_path):
    path, want = _textured_sphere(tmp_path)
    got = se3.utils.load_obj_mesh(path)
    # (vertices are numbered in order of first use by a face: compare per face corner)
    assert got["vertices"].shape == want["vertices"].shape and got["faces"].shape == want["faces"].shape
    assert np.allclose(got["vertices"][got["faces"]], want["vertices"][want["faces"]], atol=1e-8)
    assert np.allclose(got["uv"][got["faces"]], want["uv"][want["faces"]], atol=1e-8)
    assert np.array_equal(got["texture"], want["texture"]) and np.allclose(got["kd"], want["kd"])
    # quads are fan-triangulated, negative indices are relative, a vertex used with two texcoords is split
    (tmp_path / "q.obj").write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvt 0.5 0.5\n"
                                    "f 1/1 2/2 3/3 4/4\nf -4/5 -3/2 -2/3\n")
    q = se3.utils.load_obj_mesh(str(tmp_path / "q.obj"))
    assert q["faces"].tolist() == [[0, 1, 2], [0, 2, 3], [4, 1, 2]] and len(q["vertices"]) == 5 and q["texture"] is None
    lv = R.mip_pyramid(want["texture"])
    assert [l.shape[:2] for l in lv][:4] == [(64, 128), (32, 64), (16, 32), (8, 16)] and lv[-1].shape[:2] == (1, 1)
    assert abs(int(lv[-1][0, 0, 0]) - int(want["texture"][..., 0].mean())) <= 2


@pytest.mark.gpu
def test_hip_full_frame_renderer_vs_pyrender_oracle(se3, tmp_path):
    path, mesh = _textured_sphere(tmp_path, subdiv=2)
    eng = se3.Engine(0, 1)
    H, W = 120, 160
    K = np.array([[266.7, 0, 78.2], [0, 266.9, 60.3], [0, 0, 1.0]])
    for textured in (True, False):
        model = path if textured else dict(vertices=mesh["vertices"], faces=mesh["faces"],
                                           colors=R.icosphere(2, 0.05, 3)["colors"], kd=(1.0, 0.9, 0.8))
        ren = se3.HipRenderer(eng, model, mode="pyrender", frame_size=(H, W))
        P = Fx.pose(4, (0.01, -0.02, 0.45))
        rgb, depth = ren.render_frame(P, K)
        if textured:
            orgb, odepth = S.render_frame(mesh["vertices"].astype(np.float32), None, mesh["faces"], P, K, W, H,
                                          uv=mesh["uv"], texture=mesh["texture"], kd=mesh["kd"])
        else:
            orgb, odepth = S.render_frame(model["vertices"].astype(np.float32), (model["colors"] / 255.0).astype(np.float32),
                                          model["faces"], P, K, W, H, kd=model["kd"])
        assert rgb.shape == (H, W, 3) and depth.dtype == np.uint16 and depth.shape == (H, W)
        assert (depth > 0).sum() > 1500 and np.array_equal(depth, odepth)          # coverage and depth: every pixel
        d = np.abs(rgb.astype(int) - orgb.astype(int)).max(2)
        if textured:
            # the .obj loader numbers vertices by first use and the texture filter is float32 arithmetic on both sides: a texel
            # boundary can move
            assert np.median(d[depth > 0]) <= 1 and (d > 6).mean() < 0.02, (np.median(d), (d > 6).mean())
        else:
            assert np.array_equal(rgb, orgb)
        assert (rgb[~(depth > 0)] == 0).all()
        # the sphere's centre projects to (fx x/z + cx, fy y/z + cy): pixel i covers u in [i, i+1)
        ys, xs = np.nonzero(depth)
        u = K[0, 0] * P[0, 3] / P[2, 3] + K[0, 2]; v = K[1, 1] * P[1, 3] / P[2, 3] + K[1, 2]
        assert abs(xs.mean() + 0.5 - u) < 1.0 and abs(ys.mean() + 0.5 - v) < 1.0
        assert 450 - 51 <= depth[depth > 0].min() <= 450 - 45


@pytest.mark.gpu
def test_tracker_pyrenderer_route(se3, tmp_path):
    """dataset_info['renderer'] == 'pyrenderer' + a textured .obj (predict.py:161-164): render_window = crop_bbox of
    the full-frame render (predict.py:209-213), and on_track (rendered frame cropped on the device) equals the oracle
    composition fed that crop."""
    from oracle import se3_oracle as O
    path, mesh = _textured_sphere(tmp_path, subdiv=3, radius=0.06)
    sd = O.make_state_dict(0, head_gain=0.002)
    mean, std = Fx.mean_std(0)
    info = dict(Fx.DATASET_INFO); info["object_width"] = 150.0; info["renderer"] = "pyrenderer"
    trk = se3.Tracker(info, mean, std, {"state_dict": sd}, model_path=path)
    assert isinstance(trk.renderer, se3.HipRenderer) and trk.renderer.full_frame and trk.object_cloud is not None
    P0 = Fx.pose(3, (0.03, -0.01, 0.7))
    full_rgb, full_depth = trk.renderer.render_frame(P0, trk.K)
    assert full_rgb.shape == (480, 640, 3)
    rgbA, depthA = trk.render_window(P0)
    bb = O.compute_bbox(P0, Fx.K_YCB, 150.0, scale=(1000, 1000, 1000))
    wantA, wantD = O.crop_bbox(full_rgb, full_depth, bb, (176, 176))
    assert (rgbA == wantA).all() and (depthA == wantD).all() and (depthA > 100).sum() > 3000
    rgb, depth = Fx.synthetic_frame(12)
    P1 = trk.on_track(P0, rgb, depth)
    want, _ = O.on_track(sd, P0, rgb, depth, rgbA, depthA, Fx.K_YCB, 150.0, mean, std)
    assert np.abs(P1 - want).max() < 1e-5


# ---- an independent geometric pin: a finely tessellated sphere against the analytic ray-sphere intersection ----------
def _analytic_sphere_window(P, K, win, res, radius):
    """Depth (metres, camera z) a perfect sphere of `radius` at P[:3,3] has at the centre of every pixel of the
    res x res window render, and the ray discriminant (>0: the ray hits).  Window convention of predict.py:201-207:
    win = (left, top, right, bottom) in (u, cy - fy y/z); image rows count down from `bottom`."""
    left, top, right, bottom = win
    i = np.arange(res) + 0.5
    u = left + i * (right - left) / res
    vflip = bottom - i * (bottom - top) / res
    dx = (u - K[0, 2]) / K[0, 0]
    dy = (K[1, 2] - vflip) / K[1, 1]
    return _ray_sphere(dx[None, :], dy[:, None], P[:3, 3], radius)


def _analytic_sphere_frame(P, K, W, H, radius):
    """The same for the full-frame (pyrender) route: pixel (row j, column i) looks along ((i+.5-cx)/fx, (j+.5-cy)/fy, 1)."""
    dx = (np.arange(W) + 0.5 - K[0, 2]) / K[0, 0]
    dy = (np.arange(H) + 0.5 - K[1, 2]) / K[1, 1]
    return _ray_sphere(dx[None, :], dy[:, None], P[:3, 3], radius)


def _ray_sphere(dx, dy, c, radius):
    a = dx * dx + dy * dy + 1.0                      # |d|^2 with d = (dx, dy, 1): the parameter t IS the camera-space z
    b = dx * c[0] + dy * c[1] + c[2]
    disc = b * b - a * (c @ c - radius * radius)
    t = (b - np.sqrt(np.maximum(disc, 0.0))) / a
    return t, disc / (a * radius * radius)           # normalised discriminant: ~ (1 - (miss distance / radius)^2)


def _check_against_sphere(depth_mm, z, ndisc, what):
    hit = ndisc > 0
    inner = ndisc > 0.12                             # a few pixels inside the silhouette: depth is steep at the rim
    outer = ndisc < -0.12
    assert (depth_mm[inner] > 0).all(), what + ": hole inside the analytic silhouette"
    assert (depth_mm[outer] == 0).all(), what + ": coverage outside the analytic silhouette"
    area, want = int((depth_mm > 0).sum()), int(hit.sum())
    assert abs(area - want) <= 0.01 * want + 8, (what, area, want)
    err = depth_mm[inner].astype(np.float64) - z[inner] * 1000.0
    # uint16 truncation (-1 .. 0) of a surface inscribed in the sphere: the facets lie <= 0.05 mm inside it at this
    # tessellation, seen along the ray up to 1 / 0.35 times that at the edge of `inner`; GL interpolates depth from vertices
    # snapped to 1/16 pixel (<= 1/32 pixel off), which at the edge of `inner` (2 mm of depth per pixel) is +-0.06 mm
    assert err.min() > -1.0 - 0.1 and err.max() < 0.3, (what, err.min(), err.max())


def test_oracle_vs_analytic_sphere():
    """The CPU restatement of the GL pipeline (window render and full-frame render) against geometry it does not
    share any code with: projection, y-flip, window scaling, row order and depth linearisation all enter."""
    radius = 0.06
    m = R.icosphere(4, radius, 0)                    # 5120 faces
    P = np.eye(4); P[:3, 3] = (0.03, -0.02, 0.7)
    win = (250, 160, 420, 330)
    _, depth = R.render(*_mesh_args(m), P, Fx.K_YCB, win)
    z, nd = _analytic_sphere_window(P, Fx.K_YCB, win, 176, radius)
    _check_against_sphere(depth, z, nd, "oracle window render")
    H, W = 120, 160
    K = np.array([[266.7, 0, 78.2], [0, 266.9, 60.3], [0, 0, 1.0]])
    P2 = np.eye(4); P2[:3, 3] = (0.01, -0.02, 0.45)
    _, depth2 = R.render_frame(m["vertices"].astype(np.float32), (m["colors"] / 255.0).astype(np.float32), m["faces"], P2, K, W, H)
    z2, nd2 = _analytic_sphere_frame(P2, K, W, H, radius)
    _check_against_sphere(depth2, z2, nd2, "oracle full-frame render")


@pytest.mark.gpu
def test_hip_rasteriser_vs_analytic_sphere(se3):
    """The HIP rasteriser (both routes) against the analytic sphere: an oracle-independent check of K5."""
    radius = 0.06
    m = R.icosphere(5, radius, 0)                    # 20480 faces
    eng = se3.Engine(0, 1)
    P = Fx.pose(2, (0.03, -0.02, 0.7))               # rotated: the sphere's silhouette does not care
    ren = se3.HipRenderer(eng, m)
    win = se3.HipRenderer.gl_window(P, Fx.K_YCB, 150.0)
    _, depth = ren.render(P, Fx.K_YCB, win)
    z, nd = _analytic_sphere_window(P, Fx.K_YCB, win, 176, radius)
    _check_against_sphere(depth, z, nd, "HIP window render")
    H, W = 120, 160
    K = np.array([[266.7, 0, 78.2], [0, 266.9, 60.3], [0, 0, 1.0]])
    P2 = Fx.pose(3, (0.01, -0.02, 0.45))
    ren2 = se3.HipRenderer(eng, dict(vertices=m["vertices"], faces=m["faces"], colors=m["colors"], kd=(1.0, 1.0, 1.0)),
                           mode="pyrender", frame_size=(H, W))
    _, depth2 = ren2.render_frame(P2, K)
    z2, nd2 = _analytic_sphere_frame(P2, K, W, H, radius)
    _check_against_sphere(depth2, z2, nd2, "HIP full-frame render")


@pytest.mark.gpu
@pytest.mark.parametrize("subdiv,width,t", [(3, 100.0, (0.01, -0.02, 0.5)),     # the window cuts the sphere on all four sides
                                            (0, 100.0, (0.02, 0.01, 0.35)),      # 20 big triangles, most of them crossing the window
                                            (2, 150.0, (0.0, 0.0, 0.13)),        # the near plane (0.1 m) cuts the sphere
                                            (4, 90.0, (0.06, 0.05, 0.6))])       # off-centre: two sides cut, thousands of small triangles
def test_hip_rasteriser_clipping_equals_the_rules(se3, subdiv, width, t):
    """Triangles that cross the frustum take raster_queue_kernel's clip path (Sutherland-Hodgman in clip space, polygon outline
    walked edge by edge): byte equality with the statement of the GL implementation's rules, which oracle/ss_rules.py holds to the
    live library on clipped triangles (tests/test_ss_rules.py::test_clipping)."""
    from oracle import ss_fast as SF
    m = Fx.icosphere(subdiv, 0.06, 7)
    eng = se3.Engine(0, 1)
    ren = se3.HipRenderer(eng, m)
    P = Fx.pose(11, t)
    win = se3.HipRenderer.gl_window(P, Fx.K_YCB, width)
    rgb, depth = ren.render(P, Fx.K_YCB, win)
    want_rgb, want_d = SF.render_vispy(*_mesh_args(m), P, Fx.K_YCB, win, numpy_rule="numpy1")
    assert (want_d > 0).sum() > 3000
    border = np.concatenate([want_d[0], want_d[-1], want_d[:, 0], want_d[:, -1]])
    if t[2] > 0.2:
        assert (border > 0).sum() > 20                                           # the object really reaches the window's edge
    else:
        assert want_d[want_d > 0].min() <= 102                                   # ... or the near plane (100 mm)
    assert np.array_equal(depth, want_d), ((depth != want_d).sum(), np.abs(depth.astype(int) - want_d.astype(int)).max())
    assert np.array_equal(rgb, want_rgb), (rgb != want_rgb).any(2).sum()
