"""The reference's VispyRenderer under the NumPy generation the reference PINS (TEST INFRASTRUCTURE ONLY; build container).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_gl_numpy1_golden      ->  tests/golden/gl_swiftshader_numpy1.npz

vispy_renderer.py:163-169 turns the float32 depth buffer into millimetres with float64 SCALARS (`A`, `B` are elements of the
float64 projection matrix): `distance = B / (self.depth * -2.0 + 1.0 - A) * -1`.  Under every NumPy the reference can run on
(< 1.24, value-based casting) the scalars are cast to float32 and the whole expression is float32 arithmetic; under NumPy 2
(NEP 50; the interpreter that made gl_swiftshader.npz) `- A` promotes the array to float64.  The truncation to uint16 then
lands on a different millimetre for a fraction of the pixels.  Same question as OffsetDepth (oracle/make_numpy1_golden.py), same
answer: run the UNMODIFIED class under /opt/conda/bin/python3.9 (NumPy 1.26.4) on the same software GL.

  stage 1 (this interpreter): the meshes / poses of oracle/make_gl_golden.py: CASES -> .ply files + a parameter .npz;
  stage 2 (python3.9, NumPy 1.x): imports vispy_renderer.py from /root/reference through the vispy / PyOpenGL / plyfile
          stand-ins of oracle/swiftshader_gl.py (cv2, PIL.Image and `from Utils import *` are unused by the class: empty
          stubs), drives it as predict.py:193-208 does (window from stage 1: compute_bbox is integer-valued and identical), and
          stores rgb, depth and the RAW float depth buffer.
Stored: depth_<seed> (NumPy-1 millimetres), zbuf_<seed> (float32 depth buffer, rows as read), ndiff_<seed> = pixels whose
millimetre differs from the NumPy-2 golden; rgb is asserted identical to gl_swiftshader.npz."""
import os
import subprocess
import sys
import tempfile

import numpy as np

PY39 = os.environ.get("SE3TN_NUMPY1_PYTHON", "/opt/conda/bin/python3.9")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "gl_swiftshader_numpy1.npz")
REFERENCE_ROOT = os.environ.get("SE3TN_REFERENCE_ROOT", "/root/reference")


def stage2(tmp):
    import types
    assert np.__version__.startswith("1."), "stage 2 must run under NumPy 1.x, got " + np.__version__
    sys.dont_write_bytecode = True
    sys.path.insert(0, ROOT)
    from oracle import swiftshader_gl as SG
    SG.install_stubs()
    for name in ("cv2", "Utils"):
        sys.modules[name] = types.ModuleType(name)
    sys.path.insert(0, REFERENCE_ROOT)
    import vispy_renderer as VR
    assert os.path.realpath(VR.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), VR.__file__
    raw = {}
    orig = SG._glReadPixels

    def read(x, y, w, h, fmt, typ):
        buf = orig(x, y, w, h, fmt, typ)
        if fmt == SG.GL_DEPTH_COMPONENT:
            raw["z"] = np.array(buf, copy=True)
        return buf
    sys.modules["OpenGL.GL"].glReadPixels = read
    VR.gl.glReadPixels = read
    z = np.load(os.path.join(tmp, "in.npz"))
    SG.DEPTH_FORMAT["bits"] = 32
    out = {"numpy_version": np.array(np.__version__)}
    for seed in z["seeds"]:
        K = z["K"]
        ren = VR.VispyRenderer(os.path.join(tmp, "m%d.ply" % seed), K, H=176, W=176)
        left, top, right, bottom = [int(v) for v in z["window_%d" % seed]]
        ren.update_cam_mat(K, left, right, bottom, top)
        color, depth = ren.render_image(z["ob2cam_gl_%d" % seed])
        out["rgb_%d" % seed], out["depth_%d" % seed], out["zbuf_%d" % seed] = np.array(color), np.array(depth), raw["z"].reshape(176, 176)
    np.savez_compressed(os.path.join(tmp, "out.npz"), **out)


def main():
    from . import fixtures as Fx
    from .make_gl_golden import CASES, write_ply
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "gl_swiftshader.npz"))
    glcam_in_cvcam = np.diag([1.0, -1.0, -1.0, 1.0])
    with tempfile.TemporaryDirectory() as tmp:
        inp = {"seeds": np.array([c[0] for c in CASES]), "K": Fx.K_YCB}
        for seed, subdiv, t in CASES:
            write_ply(os.path.join(tmp, "m%d.ply" % seed), Fx.icosphere(subdiv, 0.05, seed))
            inp["window_%d" % seed] = g2["window_%d" % seed]
            inp["ob2cam_gl_%d" % seed] = np.linalg.inv(glcam_in_cvcam).dot(Fx.pose(seed, t))      # predict.py:205-207
        np.savez(os.path.join(tmp, "in.npz"), **inp)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONWARNINGS="ignore")
        subprocess.check_call([PY39, os.path.abspath(__file__), "--stage2", tmp], env=env)
        z = dict(np.load(os.path.join(tmp, "out.npz")))
    out = {"numpy_version": z["numpy_version"], "numpy2_version": np.array(np.__version__)}
    for seed, _, _ in CASES:
        assert np.array_equal(z["rgb_%d" % seed], g2["rgb_%d" % seed]), "the colour image does not depend on the NumPy generation"
        d1, d2 = z["depth_%d" % seed], g2["depth_%d" % seed]
        assert np.array_equal(d1 > 0, d2 > 0)
        out["depth_%d" % seed], out["zbuf_%d" % seed] = d1, z["zbuf_%d" % seed]
        out["ndiff_%d" % seed] = np.array(int((d1 != d2).sum()))
        print("case %d: %d of %d covered pixels land on another millimetre under NumPy %s (max %d)" % (
            seed, int((d1 != d2).sum()), int((d2 > 0).sum()), z["numpy_version"], int(np.abs(d1.astype(int) - d2.astype(int)).max())))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--stage2":
        stage2(sys.argv[2])
    else:
        main()
