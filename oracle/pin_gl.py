#!/usr/bin/env python3
"""Pin-on-first-contact script for the two OpenGL renderers of the reference (TEST INFRASTRUCTURE ONLY).

No OpenGL / vispy / pyrender exists in the offline build container, so the rendered image A is compared with a numpy
restatement of the GL pipeline (oracle/raster_oracle.py) and an analytic sphere only: shading and fill rules are "parity
unpinned" (DESIGN.md section 4).  Run THIS on a machine with the reference checkout and its renderer stack (vispy + PyOpenGL +
plyfile for vispy_renderer.py; pyrender + trimesh for offscreen_renderer.py):

    python oracle/pin_gl.py --reference /path/to/iros20-6d-pose-tracking [--only vispy|pyrender]
    python -m pytest tests/test_pinned_third_party.py

It drives the reference's OWN classes exactly as Tracker.render_window does (predict.py:193-215) on the seeded meshes /
poses of oracle/fixtures.py and writes tests/golden/gl_vispy.npz / gl_pyrender.npz (rgb, depth per case).  The tests skip
while the files are absent and compare the numpy oracle AND the HIP rasteriser against them once they exist."""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fixtures as Fx  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
# (mesh seed, subdivisions, translation, object_width_mm): the cases of tests/test_renderer.py::test_hip_rasteriser_vs_oracle
VISPY_CASES = [(0, 2, (0.03, -0.02, 0.65), 130.0), (1, 3, (-0.05, 0.04, 0.9), 130.0), (2, 1, (0.0, 0.0, 0.45), 130.0),
               (3, 0, (0.005, -0.004, 0.3), 130.0)]
PYR_K = np.array([[266.7, 0, 78.2], [0, 266.9, 60.3], [0, 0, 1.0]])
PYR_HW = (120, 160)


def compute_bbox(pose, K, scale_size, scale):
    """Utils.py:302-316 (numpy float64)."""
    obj = [pose[i, 3] * scale[i] for i in range(3)]
    off = scale_size / 2
    pts = np.array([[obj[0] - off, obj[1] - off, obj[2]], [obj[0] - off, obj[1] + off, obj[2]],
                    [obj[0] + off, obj[1] - off, obj[2]], [obj[0] + off, obj[1] + off, obj[2]]], np.float64)
    vus = np.zeros((4, 2))
    vus[:, 1] = pts[:, 0] * K[0, 0] / pts[:, 2] + K[0, 2]
    vus[:, 0] = pts[:, 1] * K[1, 1] / pts[:, 2] + K[1, 2]
    return np.round(vus).astype(np.int32)


def write_ply(path, m):
    v, n, c, f = m["vertices"], m["normals"], m["colors"].astype(np.uint8), m["faces"]
    with open(path, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                 "property float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n" % (len(v), len(f)))
        for i in range(len(v)):
            fh.write("%.9g %.9g %.9g %.9g %.9g %.9g %d %d %d\n" % (*v[i], *n[i], *c[i]))
        for t in f:
            fh.write("3 %d %d %d\n" % tuple(t))


def write_obj(path, m):
    """vertex-colour .obj (trimesh reads `v x y z r g b`): the untextured route of offscreen_renderer.Renderer."""
    with open(path, "w") as fh:
        for p, c in zip(m["vertices"], m["colors"] / 255.0):
            fh.write("v %.9g %.9g %.9g %.6f %.6f %.6f\n" % (*p, *c))
        for t in m["faces"]:
            fh.write("f %d %d %d\n" % tuple(t + 1))


def pin_vispy(tmp):
    from vispy_renderer import VispyRenderer          # the reference's class, unmodified
    out = {}
    glcam_in_cvcam = np.diag([1.0, -1.0, -1.0, 1.0])
    for seed, subdiv, t, width in VISPY_CASES:
        m = Fx.icosphere(subdiv, 0.05, seed)
        ply = os.path.join(tmp, "m%d.ply" % seed)
        write_ply(ply, m)
        ren = VispyRenderer(ply, Fx.K_YCB, H=176, W=176)
        P = Fx.pose(seed, t)
        # predict.py:193-208
        bbox = compute_bbox(P, Fx.K_YCB, width, (1000, -1000, 1000))
        left, right = np.min(bbox[:, 1]), np.max(bbox[:, 1])
        top, bottom = np.min(bbox[:, 0]), np.max(bbox[:, 0])
        ren.update_cam_mat(Fx.K_YCB, left, right, bottom, top)
        rgb, depth = ren.render_image(np.linalg.inv(glcam_in_cvcam).dot(P))
        out["rgb_%d" % seed], out["depth_%d" % seed] = np.array(rgb), np.array(depth)
        out["window_%d" % seed] = np.array([left, top, right, bottom])
    np.savez_compressed(os.path.join(OUT, "gl_vispy.npz"), **out)
    print("wrote gl_vispy.npz")


def pin_pyrender(tmp):
    from offscreen_renderer import Renderer           # the reference's class, unmodified
    H, W = PYR_HW
    m = Fx.icosphere(2, 0.05, 3)
    obj = os.path.join(tmp, "m.obj")
    write_obj(obj, m)
    ren = Renderer([obj], PYR_K, H, W)
    out = {}
    for i, t in enumerate([(0.01, -0.02, 0.45), (-0.03, 0.02, 0.7)]):
        P = Fx.pose(4 + i, t)
        color, depth = ren.render([P])                # predict.py:209-211
        out["rgb_%d" % i] = np.array(color)[..., :3]
        out["depth_%d" % i] = (np.array(depth) * 1000).astype(np.uint16)
    np.savez_compressed(os.path.join(OUT, "gl_pyrender.npz"), **out)
    print("wrote gl_pyrender.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of wenbowen123/iros20-6d-pose-tracking")
    ap.add_argument("--only", choices=["vispy", "pyrender"])
    a = ap.parse_args()
    sys.path.insert(0, a.reference)
    if not hasattr(np, "float"):
        np.float = float
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        if a.only in (None, "vispy"):
            pin_vispy(tmp)
        if a.only in (None, "pyrender"):
            pin_pyrender(tmp)


if __name__ == "__main__":
    main()
