"""Runs the UNMODIFIED reference renderer class `VispyRenderer` (vispy_renderer.py:47-178) on a real OpenGL implementation --
SwiftShader's OpenGL ES 3.0 through the vispy / PyOpenGL stand-ins of oracle/swiftshader_gl.py -- driven exactly as
Tracker.render_window drives it (predict.py:193-208), and stores the images as tests/golden/gl_swiftshader.npz
(TEST INFRASTRUCTURE ONLY; needs /root/reference and the kaleido wheel's SwiftShader, i.e. the build container):

    python -m oracle.make_gl_golden [out_dir]

Cases = tests/test_renderer.py::test_hip_rasteriser_vs_oracle's (mesh seed, subdivisions, translation) + two more poses; depth
attachment: GL_DEPTH_COMPONENT32F (see oracle/swiftshader_gl.py: DEPTH_FORMAT)."""
import importlib
import os
import sys
import tempfile

import numpy as np

from . import fixtures as Fx
from . import ref_shims
from . import swiftshader_gl as SG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJECT_WIDTH = 130.0
CASES = [(0, 2, (0.03, -0.02, 0.65)), (1, 3, (-0.05, 0.04, 0.9)), (2, 1, (0.0, 0.0, 0.45)), (3, 0, (0.005, -0.004, 0.3)),
         (4, 4, (0.10, 0.06, 1.2)), (5, 3, (-0.12, -0.08, 0.55))]


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


def load_reference_renderer():
    ref_shims.install()
    SG.install_stubs()
    sys.modules.pop("vispy_renderer", None)
    return importlib.import_module("vispy_renderer"), importlib.import_module("Utils")


def render_case(VR, U, tmp, seed, subdiv, t, depth_bits=16):
    """predict.py:193-208 with the reference's own compute_bbox and renderer class."""
    SG.DEPTH_FORMAT["bits"] = depth_bits
    m = Fx.icosphere(subdiv, 0.05, seed)
    ply = os.path.join(tmp, "m%d.ply" % seed)
    if not os.path.exists(ply):
        write_ply(ply, m)
    K = Fx.K_YCB
    ren = VR.VispyRenderer(ply, K, H=176, W=176)
    ob2cam = Fx.pose(seed, t)
    glcam_in_cvcam = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
    bbox = U.compute_bbox(ob2cam, K, OBJECT_WIDTH, scale=(1000, -1000, 1000))
    left = np.min(bbox[:, 1]); right = np.max(bbox[:, 1]); top = np.min(bbox[:, 0]); bottom = np.max(bbox[:, 0])
    ren.update_cam_mat(K, left, right, bottom, top)
    color, depth = ren.render_image(np.linalg.inv(glcam_in_cvcam).dot(ob2cam))
    return np.array(color), np.array(depth), np.array([left, top, right, bottom], np.int64), ren


FRAME_K = np.array([[266.7, 0, 78.2], [0, 266.9, 60.3], [0, 0, 1.0]])
FRAME_HW = (120, 160)
FRAME_POSES = [(4, (0.01, -0.02, 0.45)), (5, (-0.03, 0.02, 0.7))]
FRAME_KD_VERTEX = (1.0, 0.9, 0.8)


def render_frames():
    """The second renderer's route (offscreen_renderer.py through pyrender; not installable offline): the GL calls of that scene as
    stated in oracle/swiftshader_gl.py: render_frame_gl, on the same real GL.  Textured cases upload the ORACLE's mip pyramid level
    by level, so that only GL's sampling rules (level of detail, trilinear weights, REPEAT wrap) are compared; glGenerateMipmap's own
    pyramid is stored beside it (it differs from the 2x2 box filter with round-to-nearest by at most 2 / 255)."""
    from . import raster_oracle as R
    H, W = FRAME_HW
    ms = Fx.textured_sphere(2)
    out = {}
    levels = R.mip_pyramid(ms["texture"])
    gen = SG.read_mip_levels(ms["texture"])
    out["mipgen_max_abs_diff"] = np.array([int(np.abs(a.astype(int) - b.astype(int)).max()) for a, b in zip(gen, levels)])
    v32 = ms["vertices"].astype(np.float32)
    for i, (pseed, t) in enumerate(FRAME_POSES):
        P = Fx.pose(pseed, t)
        out["frame_tex_rgb_%d" % i], out["frame_tex_depth_%d" % i] = SG.render_frame_gl(
            v32, None, ms["faces"], P, FRAME_K, W, H, uv=ms["uv"], texture=ms["texture"], kd=ms["kd"], mip_levels=levels)
        out["frame_vc_rgb_%d" % i], out["frame_vc_depth_%d" % i] = SG.render_frame_gl(
            v32, (ms["colors"] / 255.0).astype(np.float32), ms["faces"], P, FRAME_K, W, H, kd=FRAME_KD_VERTEX)
    return out


def main(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    VR, U = load_reference_renderer()
    gl = SG.GL.get()
    out = {"gl_version": gl.version, "gl_renderer": gl.renderer}
    with tempfile.TemporaryDirectory() as tmp:
        for seed, subdiv, t in CASES:
            for bits in (32,):
                rgb, depth, win, ren = render_case(VR, U, tmp, seed, subdiv, t, bits)
                sfx = ""
                out["rgb_%d%s" % (seed, sfx)], out["depth_%d%s" % (seed, sfx)] = rgb, depth
                out["window_%d" % seed] = win
            print("case %d: %d covered pixels, depth %d..%d mm" % (seed, int((depth > 0).sum()), int(depth[depth > 0].min()), int(depth.max())))
    out.update(render_frames())
    np.savez_compressed(os.path.join(out_dir, "gl_swiftshader.npz"), **out)
    print("wrote gl_swiftshader.npz (%s, %s)" % (gl.version, gl.renderer))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
