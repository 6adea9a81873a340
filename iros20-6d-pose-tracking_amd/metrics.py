"""Pose-error metrics the reference reports its results with -- CPU, as in the
reference.  `add` / `adi`: Utils.py:72-98 (ADD / ADD-S, Hinterstoisser et al.);  `VOCap`:
eval_ycb.py:45-64 (area under the accuracy-threshold curve up to 0.1 m).
`adi` uses cKDTree.query(workers=...) -- the reference's `n_jobs=10` (Utils.py:96) raises TypeError on
SciPy >= 1.6."""
import numpy as np
from scipy import spatial


def _points(model):
    return np.asarray(model.points if hasattr(model, "points") else model, np.float64)


def _transform(pts, T):
    return pts @ np.asarray(T)[:3, :3].T + np.asarray(T)[:3, 3]


def add(pred, gt, model):
    """Utils.py:72-82: mean distance between corresponding model points under the two poses."""
    pts = _points(model)
    return float(np.linalg.norm(_transform(pts, pred) - _transform(pts, gt), axis=1).mean())


def adi(pred, gt, model, workers=-1):
    """Utils.py:84-98: mean closest-point distance (symmetric objects)."""
    pts = _points(model)
    tree = spatial.cKDTree(_transform(pts, pred))
    d, _ = tree.query(_transform(pts, gt), k=1, workers=workers)
    return float(d.mean())


def VOCap(rec):
    """eval_ycb.py:45-64.  Raises IndexError when no error is below 0.1 (as the reference does)."""
    rec = np.sort(np.array(rec))
    n = len(rec)
    prec = np.arange(1, n + 1) / float(n)
    rec = rec.reshape(-1)
    prec = prec.reshape(-1)
    index = np.where(rec < 0.1)[0]
    rec = rec[index]
    prec = prec[index]
    mrec = [0, *list(rec), 0.1]
    mpre = [0, *list(prec), prec[-1]]
    for i in range(1, len(mpre)):
        mpre[i] = max(mpre[i], mpre[i - 1])
    mpre = np.array(mpre)
    mrec = np.array(mrec)
    i = np.where(mrec[1:] != mrec[0:len(mrec) - 1])[0] + 1
    return float(np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10)
