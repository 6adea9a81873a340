"""The reference's OffsetDepth / NormalizeChannels / ToTensor under the NumPy generation the reference PINS.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_numpy1_golden      ->  tests/golden/preprocess_numpy1.npz

TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference and /opt/conda/bin/python3.9).

Why: `depth -= pose[2,3]*1000` (data_augmentation.py:134-144) subtracts a float64 SCALAR from a float32
array.  The reference pins Python 3.6 (docker/dockerfile:28) => NumPy <= 1.19, and its own `np.float` (Utils.py:307) stops
importing at NumPy 1.24: every NumPy the reference can run on uses value-based casting, under which the scalar is cast to
float32 FIRST and the subtraction is a float32 operation.  NumPy 2 (NEP 50: this image's main interpreter, the one that made
tests/golden/preprocess.npz) keeps the float64 scalar, computes in float64 and rounds once: <= 1 ulp(f32) different.
The image also carries /opt/conda/bin/python3.9 with NumPy 1.26.4 (value-based casting, like every 1.x): this script runs the
UNMODIFIED reference classes there.

Two stages, because torch / this repo's fixtures need the main interpreter and NumPy 1.x needs the other:
  stage 1 (this interpreter): the five PRE_CASES of oracle/make_golden.py -> rendered A, cropped B (integer work, the
          reference's own crop_bbox through ref_shims), pose, mean / std -> a temporary .npz;
  stage 2 (`/opt/conda/bin/python3.9 oracle/make_numpy1_golden.py --stage2 in.npz out.npz`, NumPy 1.26.4): imports the
          reference's data_augmentation.py with stub modules for the imports these three classes do not use (cv2, Utils,
          torch -- `torch.from_numpy` is only a container change at data_augmentation.py:188-189) and runs
          OffsetDepth -> NormalizeChannels -> ToTensor exactly as TrackDataset.processData does (datasets.py:135-136).
Stored per case: sha256 of dataA / dataB (float32 [4,176,176]), a sub-sample, and the number of elements that differ from
the NumPy-2 tensors + the largest difference (the statistics the DESIGN quotes)."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

PY39 = os.environ.get("SE3TN_NUMPY1_PYTHON", "/opt/conda/bin/python3.9")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "preprocess_numpy1.npz")
REFERENCE_ROOT = os.environ.get("SE3TN_REFERENCE_ROOT", "/root/reference")
SUB = 7
# Beyond PRE_CASES (whose z are whole millimetres: 800.0, 750.0 ... are exact in float32, so both rules agree on them bit for bit):
# poses whose z * 1000 is NOT a float32 -- the only place the two NumPy generations differ.  (name, seed, z [m]); rendered-size
# 176 x 176 images on both sides (no crop involved), incl. the `pose[2,3] < 0` branch of data_augmentation.py:137-138.
OFFSET_CASES = [("fracz", 21, 0.8123456789), ("fracz_far", 22, 1.2345678901), ("fracz_near", 23, 0.4567890123),
                ("fracz_gl", 24, -0.7654321098)]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def stage2(inp, outp):
    import types
    assert np.__version__.startswith("1."), "stage 2 must run under NumPy 1.x, got " + np.__version__

    class _T:   # what `torch.from_numpy(x)` has to be for ToTensor (data_augmentation.py:188-189): a holder
        def __init__(self, a):
            self.a = a

        def numpy(self):
            return self.a

    torch = types.ModuleType("torch")
    torch.from_numpy = lambda a: _T(a)
    sys.modules["torch"] = torch
    sys.modules["cv2"] = types.ModuleType("cv2")
    sys.modules["Utils"] = types.ModuleType("Utils")      # `from Utils import *`: nothing of it is used by the three classes
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE_ROOT)
    import data_augmentation as DA
    assert os.path.realpath(DA.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), DA.__file__

    z = np.load(inp)
    names = [str(s) for s in z["names"]]
    mean, std = z["mean"], z["std"]
    post = [DA.OffsetDepth(), DA.NormalizeChannels(mean, std), DA.ToTensor()]
    d = {"numpy_version": np.array(np.__version__), "names": z["names"]}
    for name in names:
        rgbA, depthA, rgbB, depthB, P = (z[name + "_" + k] for k in ("rgbA", "depthA", "rgbB", "depthB", "pose"))
        # datasets.py:121-123,135-136: masks from depth > 100, then the posttransforms on the 7-tuple
        data = (rgbA, depthA, rgbB, depthB, depthA > 100, depthB > 100, P)
        for t in post:
            data = t(data)
        a, b = data[0][0].numpy(), data[0][1].numpy()
        assert a.dtype == np.float32 and a.shape == (4, 176, 176) and b.shape == (4, 176, 176)
        d[name + "_dataA_sha"] = np.array(_sha(a)); d[name + "_dataB_sha"] = np.array(_sha(b))
        d[name + "_dataA_sub"] = a[:, ::SUB, ::SUB].copy(); d[name + "_dataB_sub"] = b[:, ::SUB, ::SUB].copy()
        d[name + "_dataA_depth"] = a[3].copy(); d[name + "_dataB_depth"] = b[3].copy()   # stage 1 compares, then drops them
    np.savez_compressed(outp, **d)


def main():
    from . import fixtures as Fx
    from . import ref_shims
    from . import se3_oracle as O
    from .make_golden import PRE_CASES
    assert os.path.exists(PY39), PY39 + " (the NumPy 1.x interpreter of this image) is missing"
    ref = ref_shims.load()
    mean, std = Fx.mean_std(0)
    inp = {"names": np.array([c[0] for c in PRE_CASES]), "mean": mean, "std": std}
    np2 = {}
    for name, fseed, t, width in PRE_CASES:
        rgb, depth = Fx.synthetic_frame(fseed)
        P = Fx.pose(fseed, t)
        rgbA, depthA = Fx.synthetic_render(fseed + 100, t[2])
        bb = ref.Utils.compute_bbox(P, Fx.K_YCB, width, scale=(1000, 1000, 1000))
        rgbB, depthB = ref.Utils.crop_bbox(rgb, depth, bb, (176, 176))
        inp.update({name + "_rgbA": rgbA, name + "_depthA": depthA, name + "_rgbB": rgbB, name + "_depthB": depthB, name + "_pose": P})
        np2[name] = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std, offset_rule="numpy2")
    for name, seed, zz in OFFSET_CASES:
        P = Fx.pose(seed, (0.02, -0.01, zz))
        rgbA, depthA = Fx.synthetic_render(seed + 100, abs(zz))
        rgbB, depthB = Fx.synthetic_render(seed + 200, abs(zz))
        inp.update({name + "_rgbA": rgbA, name + "_depthA": depthA, name + "_rgbB": rgbB, name + "_depthB": depthB, name + "_pose": P})
        np2[name] = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std, offset_rule="numpy2")
        # the NumPy-2 side of these cases from the reference's own classes in THIS interpreter (as make_golden.gold_preprocess does)
        DA = ref.data_augmentation
        data = (rgbA, depthA, rgbB, depthB, depthA > 100, depthB > 100, P)
        for t_ in (DA.OffsetDepth(), DA.NormalizeChannels(mean, std), DA.ToTensor()):
            data = t_(data)
        ra, rb = data[0][0].numpy(), data[0][1].numpy()
        assert _sha(ra) == _sha(np2[name][0]) and _sha(rb) == _sha(np2[name][1]), "oracle numpy2 rule != the reference under NumPy 2"
        inp[name + "_numpy2_shaA"] = np.array(_sha(ra)); inp[name + "_numpy2_shaB"] = np.array(_sha(rb))
    inp["names"] = np.array([c[0] for c in PRE_CASES] + [c[0] for c in OFFSET_CASES])
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.npz"), os.path.join(td, "out.npz")
        np.savez(fin, **inp)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONWARNINGS="ignore")
        subprocess.check_call([PY39, os.path.abspath(__file__), "--stage2", fin, fout], env=env)
        z = dict(np.load(fout))
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "preprocess.npz"))
    out = {k: v for k, v in z.items() if not k.endswith("_depth")}
    out["numpy2_version"] = np.array(np.__version__)
    for name in [str(s_) for s_ in inp["names"]]:
        a2, b2 = np2[name]
        if name + "_dataA_sha" in g2:
            assert _sha(a2) == str(g2[name + "_dataA_sha"]) and _sha(b2) == str(g2[name + "_dataB_sha"])   # same inputs as preprocess.npz
        out[name + "_dataA_sha_numpy2"] = np.array(_sha(a2)); out[name + "_dataB_sha_numpy2"] = np.array(_sha(b2))
        out[name + "_pose"] = inp[name + "_pose"]
        for which, x2 in (("A", a2), ("B", b2)):
            x1 = z[name + "_data%s_depth" % which]
            diff = np.abs(x1.astype(np.float64) - x2[3].astype(np.float64))
            out[name + "_data%s_ndiff_vs_numpy2" % which] = np.array(int((x1 != x2[3]).sum()))
            out[name + "_data%s_maxdiff_vs_numpy2" % which] = np.array(float(diff.max()))
            # the oracle's statement of the NumPy 1.x rule reproduces what NumPy 1.x did, bit for bit
            a1, b1 = O.process_data(inp[name + "_rgbA"], inp[name + "_depthA"], inp[name + "_pose"], inp[name + "_rgbB"],
                                    inp[name + "_depthB"], mean, std, offset_rule="numpy1")
            assert _sha(a1) == str(z[name + "_dataA_sha"]) and _sha(b1) == str(z[name + "_dataB_sha"]), name
        print("%-9s  NumPy %s vs NumPy %s: %d / %d normalised depth values differ (A), %d (B), max |d| %.3e" % (
            name, str(z["numpy_version"]), np.__version__, int(out[name + "_dataA_ndiff_vs_numpy2"]), 176 * 176,
            int(out[name + "_dataB_ndiff_vs_numpy2"]),
            max(float(out[name + "_dataA_maxdiff_vs_numpy2"]), float(out[name + "_dataB_maxdiff_vs_numpy2"]))))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--stage2":
        stage2(sys.argv[2], sys.argv[3])
    else:
        sys.exit(main())
