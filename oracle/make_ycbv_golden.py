"""The reference's YCB-Video drivers, end to end and UNMODIFIED (TEST INFRASTRUCTURE ONLY; build container):

    python -m oracle.make_ycbv_golden [out_dir]        ->  tests/golden/driver_ycbv.npz

  * `predict.predictSequenceYcb()` (predict.py:446-575; BASELINE configs[2]'s driver) on sequence 0048 of the synthetic tree of
    oracle/ycbv_fixtures.py with `--reinit_frames 0048/000005,0048/000008` (PoseCNN re-initialisation through
    `use_posecnn_res`, :88-123, :538-541), its result files %05d.txt / %05dgt.txt and the ADD-S AUC it prints;
  * `predict.getResultsYcb()` (:299-443) over the tree (test sequences 48..59 that contain the class), its seq<ID>/%07d.txt files;
  * `eval_ycb.eval_one_class(args)` (eval_ycb.py:67-119) on those files: sorted ADD-S / ADD errors and both AUCs;
  * `predict.use_posecnn_res(class_id, 'SSSS/FFFFFF')` for a list of query frames (keyframes and non-keyframes: the neighbour
    search), the 4x4 it returns.
Recorded by a pass-through wrapper around Tracker.render_window: the pose fed in and the image A of every on_track call.

The reference's Tracker / renderer / network run as in oracle/make_predict_golden.py (torch-CPU, VispyRenderer on SwiftShader).
What this script supplies AROUND the unmodified functions -- none of it is on the arithmetic path:
  * the module globals predict.py's __main__ block sets (args, dataset_info, images_mean / images_std, ckpt_dir, model_path, outdir);
  * `predict.ycb_dir`: use_posecnn_res reads the GLOBAL `ycb_dir` (:90), which the file never defines -- as published,
    `--reinit_frames` ends in a NameError; the evident intent (args.ycb_dir) is supplied;
  * path remaps for two hard-coded absolute paths: `scipy.io.loadmat('/YCB_Video_toolbox/results_PoseCNN_RSS2018/...')` (:114: the
    `.format(args.ycb_dir)` has no placeholder, so the path is absolute) -> <ycb>/YCB_Video_toolbox/..., and the author's
    dataset directory inside Utils.findClassContainedVideosYcb (Utils.py:109) -> <ycb>/data_organized/;
  * predictSequenceYcb reads `<args.ycb_dir>/%04d/` (:450) while getResultsYcb / eval_ycb read `<args.ycb_dir>/data_organized/%04d/`:
    the first is run with args.ycb_dir = <ycb>/data_organized;
  * `transformations.quaternion_matrix` (C. Gohlke's transformations.py, not installable offline; unpinned third-party rule like
    cv2.Rodrigues): restated below from its published algorithm;
  * GUI / video stand-ins: cv2.VideoWriter, imread, circle, putText, imshow, waitKey, cvtColor, resize of the visualisation."""
import contextlib
import glob as _glob
import io
import math
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

from . import fixtures as Fx
from . import se3_oracle as O
from . import ycbv_fixtures as YF
from .make_gl_golden import write_ply
from .make_predict_golden import HEAD_GAIN, MESH, OBJECT_WIDTH, load_predict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARD_DATASET = "/media/bowen/e25c9489-2f57-42dd-b076-021c59369fec/DATASET/Tracking/YCB_Video_Dataset/data_organized/"
POSECNN_QUERIES = ("0048/000001", "0048/000002", "0048/000003", "0048/000005", "0048/000006", "0048/000008", "0050/000002", "0050/000004")


def quaternion_matrix(quaternion):
    """transformations.quaternion_matrix: 4x4 from (w, x, y, z); quaternions shorter than 4 eps give the identity"""
    q = np.array(quaternion, dtype=np.float64, copy=True)
    n = np.dot(q, q)
    if n < np.finfo(float).eps * 4.0:
        return np.identity(4)
    q *= math.sqrt(2.0 / n)
    q = np.outer(q, q)
    return np.array([[1.0 - q[2, 2] - q[3, 3], q[1, 2] - q[3, 0], q[1, 3] + q[2, 0], 0.0],
                     [q[1, 2] + q[3, 0], 1.0 - q[1, 1] - q[3, 3], q[2, 3] - q[1, 0], 0.0],
                     [q[1, 3] - q[2, 0], q[2, 3] + q[1, 0], 1.0 - q[1, 1] - q[2, 2], 0.0],
                     [0.0, 0.0, 0.0, 1.0]])


def _install_driver_stubs(predict, ycb):
    cv2 = sys.modules["cv2"]

    def imread(path, flags=1):
        a = np.array(Image.open(path))
        return a if flags == cv2.IMREAD_UNCHANGED or a.ndim == 2 else a[..., ::-1].copy()
    cv2.imread = imread
    cv2.circle = cv2.putText = cv2.imwrite = lambda *a, **k: None
    cv2.FONT_HERSHEY_SIMPLEX = 0
    cv2.VideoWriter_fourcc = lambda *a: 0

    class VideoWriter:
        def __init__(self, *a, **k):
            pass

        def write(self, frame):
            pass

        def release(self):
            pass
    cv2.VideoWriter = VideoWriter
    nearest = cv2.resize
    if not getattr(nearest, "_viz_ok", False):
        def resize(img, dsize, interpolation=None, **k):
            if interpolation == cv2.INTER_NEAREST:
                return nearest(img, dsize, interpolation=cv2.INTER_NEAREST)
            return np.zeros(dsize[::-1] + (3,), np.uint8)              # the half-size visualisation frame (unused result)
        resize._viz_ok = True
        cv2.resize = resize
    sys.modules["transformations"].quaternion_matrix = quaternion_matrix
    predict.T.quaternion_matrix = quaternion_matrix
    import scipy.io
    import scipy.spatial as spatial
    if not getattr(spatial.cKDTree, "_se3tn_njobs", False):          # Utils.adi: query(n_jobs=) predates SciPy 1.6
        base = spatial.cKDTree

        class _Tree(base):
            _se3tn_njobs = True

            def query(self, x, k=1, eps=0, p=2, distance_upper_bound=np.inf, n_jobs=None, workers=1):
                return base.query(self, x, k=k, eps=eps, p=p, distance_upper_bound=distance_upper_bound, workers=n_jobs or workers)
        _Tree.__name__ = "cKDTree"
        spatial.cKDTree = _Tree
    real_loadmat, real_glob = scipy.io.loadmat, _glob.glob

    def loadmat(path, *a, **k):
        if path.startswith("/YCB_Video_toolbox/"):
            path = ycb + path
        return real_loadmat(path, *a, **k)

    def glob(pattern, *a, **k):
        if pattern.startswith(HARD_DATASET):
            pattern = os.path.join(ycb, "data_organized") + "/" + pattern[len(HARD_DATASET):]
        return real_glob(pattern, *a, **k)
    scipy.io.loadmat, _glob.glob = loadmat, glob
    return lambda: (setattr(scipy.io, "loadmat", real_loadmat), setattr(_glob, "glob", real_glob))


def run(tmp):
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    predict = load_predict()
    ycb = YF.make_tree(tmp)
    restore = _install_driver_stubs(predict, ycb)
    mean, std = Fx.mean_std(0)
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    ckpt = os.path.join(tmp, "model_best_val.pth.tar")
    torch.save({"state_dict": sd}, ckpt)
    ply = os.path.join(tmp, "model.ply")
    write_ply(ply, Fx.icosphere(*MESH))
    predict.dataset_info = dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH)
    predict.images_mean, predict.images_std = mean, std
    predict.ckpt_dir, predict.model_path = ckpt, ply
    predict.ycb_dir = ycb                                              # the global use_posecnn_res reads (see the docstring)
    rec = {"rgbA": [], "depthA": [], "poses_in": []}
    orig = predict.Tracker.render_window

    def recording(self, ob2cam):
        rgb, depth = orig(self, ob2cam)
        rec["rgbA"].append(np.array(rgb)); rec["depthA"].append(np.array(depth)); rec["poses_in"].append(np.array(ob2cam))
        return rgb, depth
    predict.Tracker.render_window = recording
    out = {}
    try:
        # ---- predictSequenceYcb: sequence 0048, GT initialisation, PoseCNN re-initialisation at two frames ----------------
        outdir = os.path.join(tmp, "out_ycbv") + "/"
        predict.outdir = outdir
        predict.args = types.SimpleNamespace(ycb_dir=os.path.join(ycb, "data_organized"), seq_id=48, class_id=YF.CLASS_ID,
                                             reinit_frames=YF.REINIT_FRAMES)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            predict.predictSequenceYcb()
        m = re.search(r"adi_auc ([0-9.eE+-]+)", buf.getvalue())
        files = sorted(_glob.glob(outdir + "[0-9][0-9][0-9][0-9][0-9].txt"))
        out["ycbv_files"] = np.array([os.path.basename(f) for f in files])
        out["ycbv_poses"] = np.array([np.loadtxt(f) for f in files])
        out["ycbv_gt"] = np.array([np.loadtxt(f[:-4] + "gt.txt") for f in files])
        out["ycbv_adi_auc"] = np.float64(m.group(1))
        out["ycbv_reinit_log"] = np.array(re.findall(r"Reinitialized at\s+(\d+)", buf.getvalue()), dtype=np.int64)
        n = len(files) - 1
        assert len(rec["rgbA"]) == 2 * n                                  # image A + the visualisation render, per frame
        out["ycbv_rgbA"], out["ycbv_depthA"] = np.array(rec["rgbA"][0::2]), np.array(rec["depthA"][0::2])
        out["ycbv_poses_in"] = np.array(rec["poses_in"][0::2])
        for k in rec:
            rec[k].clear()
        # ---- getResultsYcb: every test sequence that contains the class --------------------------------------------------
        resdir = os.path.join(tmp, "res") + "/"
        predict.outdir = resdir
        predict.args = types.SimpleNamespace(ycb_dir=ycb, class_id=YF.CLASS_ID, reinit_frames=None)
        with contextlib.redirect_stdout(io.StringIO()):
            predict.getResultsYcb()
        files = sorted(_glob.glob(resdir + "seq*/*.txt"))
        out["res_files"] = np.array([os.path.relpath(f, resdir) for f in files])
        out["res_poses"] = np.array([np.loadtxt(f) for f in files])
        out["res_rgbA"], out["res_depthA"] = np.array(rec["rgbA"][0::2]), np.array(rec["depthA"][0::2])
        out["res_poses_in"] = np.array(rec["poses_in"][0::2])
        # ---- eval_ycb.eval_one_class on what getResultsYcb wrote ------------------------------------------------------------
        ey = sys.modules["eval_ycb"]
        with contextlib.redirect_stdout(io.StringIO()):
            adi_errs, add_errs = ey.eval_one_class(types.SimpleNamespace(res_dir=resdir, ycb_dir=ycb, class_id=YF.CLASS_ID))
        out["eval_adi_errs"], out["eval_add_errs"] = np.asarray(adi_errs), np.asarray(add_errs)
        out["eval_adi_auc"], out["eval_add_auc"] = np.float64(ey.VOCap(adi_errs) * 100), np.float64(ey.VOCap(add_errs) * 100)
        # ---- use_posecnn_res ----------------------------------------------------------------------------------------------------
        predict.args = types.SimpleNamespace(ycb_dir=ycb)
        with contextlib.redirect_stdout(io.StringIO()):
            out["posecnn_poses"] = np.array([predict.use_posecnn_res(YF.CLASS_ID, q) for q in POSECNN_QUERIES])
        out["posecnn_queries"] = np.array(POSECNN_QUERIES)
    finally:
        predict.Tracker.render_window = orig
        restore()
    return out


def main(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as tmp:
        g = run(tmp)
    np.savez_compressed(os.path.join(out_dir, "driver_ycbv.npz"), **g)
    print("driver_ycbv.npz: predictSequenceYcb %d files (reinit logged at %s, adi_auc %.4f); getResultsYcb %d files %s .. %s; "
          "eval_one_class %d keyframes, ADD-S AUC %.4f, ADD AUC %.4f" % (
              len(g["ycbv_files"]), g["ycbv_reinit_log"].tolist(), float(g["ycbv_adi_auc"]), len(g["res_files"]), g["res_files"][0],
              g["res_files"][-1], len(g["eval_adi_errs"]), float(g["eval_adi_auc"]), float(g["eval_add_auc"])))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
