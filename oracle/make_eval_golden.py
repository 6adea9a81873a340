"""Runs the REFERENCE's own evaluators on the synthetic trees of oracle/eval_fixtures.py and stores what they compute in
tests/golden/eval_reference.npz (TEST INFRASTRUCTURE ONLY; needs /root/reference, so it runs in the build container only):

    python -m oracle.make_eval_golden [out_dir]

  * eval_ycb.eval_one_class(args)       (eval_ycb.py:67-119): sorted ADD-S / ADD error arrays + VOCap x 100 of each
  * eval_ycbineoat.eval_all(args)       (eval_ycbineoat.py:49-109): per-object and overall ADD-S / ADD AUC -- the function
                                        only prints them, so its VOCap is wrapped to record every (errors, value) pair

The reference modules are imported UNMODIFIED through oracle/ref_shims.py plus, here:
  * an `open3d` stand-in with exactly what Utils.toOpen3dCloud / add / adi touch (geometry.PointCloud with .points,
    .colors, .transform; utility.Vector3dVector),
  * cKDTree.query(n_jobs=) -> workers= (Utils.py:96 predates SciPy 1.6),
  * `U` injected into eval_ycbineoat's namespace: the file calls U.add / U.adi / U.toOpen3dCloud but only does
    `from Utils import *` (a NameError upstream as published)."""
import importlib
import os
import sys
import tempfile
import types

import numpy as np

from . import eval_fixtures as EF
from . import ref_shims

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASS_ID = 2


def _open3d_stub():
    o3d = types.ModuleType("open3d")

    class PointCloud:
        def __init__(self):
            self.points = np.zeros((0, 3)); self.colors = np.zeros((0, 3))

        def transform(self, T):
            T = np.asarray(T, np.float64)
            self.points = np.asarray(self.points) @ T[:3, :3].T + T[:3, 3]
            return self
    o3d.geometry = types.SimpleNamespace(PointCloud=PointCloud)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.array(a, np.float64))
    return o3d


def load_reference_evaluators():
    ref_shims.install()
    sys.modules["open3d"] = _open3d_stub()
    try:
        import matplotlib.pyplot  # noqa: F401  (eval_ycb imports pyplot at module level without using it)
    except Exception:   # noqa: BLE001
        for m in ("matplotlib", "matplotlib.pyplot"):
            sys.modules.setdefault(m, types.ModuleType(m))
    import scipy.spatial as spatial
    if not getattr(spatial.cKDTree, "_se3tn_njobs", False):
        base = spatial.cKDTree

        class _Tree(base):   # the extension type is immutable: subclass and rebind the name Utils.adi looks up
            _se3tn_njobs = True

            def query(self, x, k=1, eps=0, p=2, distance_upper_bound=np.inf, n_jobs=None, workers=1):
                return base.query(self, x, k=k, eps=eps, p=p, distance_upper_bound=distance_upper_bound, workers=n_jobs or workers)
        _Tree.__name__ = "cKDTree"
        spatial.cKDTree = _Tree
    for name in ("Utils", "eval_ycb", "eval_ycbineoat"):
        sys.modules.pop(name, None)
    U = importlib.import_module("Utils")
    ey = importlib.import_module("eval_ycb")
    ee = importlib.import_module("eval_ycbineoat")
    ee.U = U
    return U, ey, ee


def run_reference(tmp, sequence_module):
    """Builds the trees under `tmp`, runs both reference evaluators, returns the dict that is stored as the golden."""
    U, ey, ee = load_reference_evaluators()
    ycb = EF.make_ycb_tree(tmp, CLASS_ID)
    res = os.path.join(tmp, "res_ycb") + "/"
    EF.make_ycb_results(sequence_module, ycb, res, CLASS_ID)
    args = types.SimpleNamespace(res_dir=res, ycb_dir=ycb, class_id=CLASS_ID)
    adi_errs, add_errs = ey.eval_one_class(args)
    out = {"ycb_adi_errs": np.asarray(adi_errs), "ycb_add_errs": np.asarray(add_errs),
           "ycb_adi_auc": ey.VOCap(adi_errs) * 100, "ycb_add_auc": ey.VOCap(add_errs) * 100}
    data = EF.make_eoat_tree(tmp)
    res2 = os.path.join(tmp, "res_eoat") + "/"
    EF.make_eoat_results(sequence_module, data, res2)
    calls = []
    orig = ee.VOCap

    def recording(rec):
        v = orig(rec)
        calls.append((np.sort(np.asarray(rec, np.float64)), v * 100))
        return v
    ee.VOCap = recording
    try:
        ee.eval_all(types.SimpleNamespace(res_dir=res2, YCBInEOAT_dir=data, ycb_dir=ycb, class_id=1))
    finally:
        ee.VOCap = orig
    objects = ["cracker", "bleach", "sugar", "tomato", "mustard"]        # eval_ycbineoat.py:48, the order of its loop
    assert len(calls) == 2 * len(objects) + 2
    for i, o in enumerate(objects):
        out["eoat_%s_adi_errs" % o], out["eoat_%s_adi_auc" % o] = calls[2 * i]
        out["eoat_%s_add_errs" % o], out["eoat_%s_add_auc" % o] = calls[2 * i + 1]
    out["eoat_all_adi_errs"], out["eoat_all_adi_auc"] = calls[-2]
    out["eoat_all_add_errs"], out["eoat_all_add_auc"] = calls[-1]
    return out


def main(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    seq = importlib.import_module("iros20-6d-pose-tracking_amd.sequence")
    with tempfile.TemporaryDirectory() as tmp:
        g = run_reference(tmp, seq)
    np.savez_compressed(os.path.join(out_dir, "eval_reference.npz"), **g)
    print("eval_reference.npz: ycb n=%d adi_auc=%.6f add_auc=%.6f | eoat n=%d adi_auc=%.6f add_auc=%.6f" % (
        len(g["ycb_adi_errs"]), g["ycb_adi_auc"], g["ycb_add_auc"], len(g["eoat_all_adi_errs"]), g["eoat_all_adi_auc"],
        g["eoat_all_add_auc"]))


if __name__ == "__main__":
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    main(sys.argv[1] if len(sys.argv) > 1 else None)
