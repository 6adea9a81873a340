"""Rasteriser: the HIP kernels against the CPU restatement of the reference's
OpenGL pipeline (oracle/raster_oracle.py; parity with a real GL driver is unpinned), plus properties."""
import numpy as np
import pytest
import torch

from oracle import fixtures as Fx
from oracle import raster_oracle as R


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
@pytest.mark.parametrize("seed,subdiv,t", [(0, 2, (0.03, -0.02, 0.65)), (1, 3, (-0.05, 0.04, 0.9)), (2, 1, (0.0, 0.0, 0.45))])
def test_hip_rasteriser_vs_oracle(se3, seed, subdiv, t):
    m = R.icosphere(subdiv, 0.05, seed)
    eng = se3.Engine(0, 1)
    ren = se3.HipRenderer(eng, m)
    P = Fx.pose(seed, t)
    win = se3.HipRenderer.gl_window(P, Fx.K_YCB, 130.0)
    rgb, depth = ren.render(P, Fx.K_YCB, win)
    orgb, odepth = R.render(*_mesh_args(m), P, Fx.K_YCB, win)
    assert rgb.shape == (176, 176, 3) and depth.dtype == np.uint16
    cover_same = (depth > 0) == (odepth > 0)
    assert cover_same.mean() > 0.9995                      # identical coverage up to float ties on edges
    both = (depth > 0) & (odepth > 0)
    assert both.sum() > 2000
    assert np.abs(depth[both].astype(int) - odepth[both].astype(int)).max() <= 1
    drgb = np.abs(rgb[both].astype(int) - orgb[both].astype(int))
    assert drgb.max() <= 2 and (drgb > 0).mean() < 0.02
    assert (rgb[~(depth > 0)] == 0).all()                  # background exactly 0 (Tracker's maskA = depthA > 100)
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
