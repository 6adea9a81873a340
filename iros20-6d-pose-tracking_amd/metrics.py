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


def VOCap(errors, max_err=0.1):
    """Area under the accuracy-vs-error-threshold curve for thresholds in [0, max_err], scaled so that a
    perfect result gives 1 (the YCB-Video toolbox metric the reference reports x100; same numbers as
    eval_ycb.py:45-64).  accuracy(t) = fraction of ALL errors <= t, a step function that jumps at every
    error below max_err; the area is summed as (gap between consecutive jump positions) x (accuracy after
    the jump), the last gap running up to max_err.  Ties contribute zero-width gaps, so they need no
    special casing.  Raises IndexError when no error is below max_err, as the reference does (a lost track);
    `auc()` maps that case to 0."""
    e = np.sort(np.asarray(errors, np.float64).ravel())
    k = int(np.searchsorted(e, max_err, side="left"))        # errors strictly below the cap
    if k == 0:
        raise IndexError("VOCap: no error below %g" % max_err)
    x = np.concatenate(([0.0], e[:k], [max_err]))
    acc = np.concatenate((np.arange(1, k + 1, dtype=np.float64) / e.size, [k / float(e.size)]))
    return float(np.dot(np.diff(x), acc) / max_err)


def auc(errors, max_err=0.1):
    """VOCap x 100 with the lost-track / empty case reported as 0.0 instead of an exception."""
    try:
        return VOCap(errors, max_err) * 100.0
    except IndexError:
        return 0.0
