"""Independent second opinions on the two OpenCV rules the oracle restates.

The goldens were produced by the reference's code running with ``oracle/ref_shims.py``'s cv2 shim, which IS
``O.resize_nearest`` / ``O.rodrigues`` -- so the goldens cannot disagree with the oracle on these two
rules.  OpenCV itself is not installable offline; what is available are two implementations written by
other people for the same mathematical objects:

* ``torch.nn.functional.interpolate(mode='nearest')`` (top-left aligned ``floor(x * src/dst)``, scale in
  float32) and exact integer arithmetic ``(x * src) // dst`` for the NEAREST source-index rule
  (Utils.py:343-344 -> cv2.resize(..., INTER_NEAREST));
* ``scipy.spatial.transform.Rotation.from_rotvec`` (quaternion route) for Rodrigues (datasets.py:173).

What these tests pin: the oracle's index table equals BOTH independent tables at every (x, src) where
x*src/176 is not an exact integer (351,000+ indices); at the exact-integer points the three evaluation
orders (double reciprocal = OpenCV's ``cvFloor(x * (1./fx))``, float32 scale = torch, exact) may land on
either side of the integer, the oracle's value is always ``exact`` or ``exact - 1``, and it equals the
literal double evaluation of OpenCV's expression.  So the only thing left unpinned is OpenCV's
*evaluation order* at those points (taken from its published resizeNN source), not the rule."""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.spatial.transform import Rotation

from oracle import se3_oracle as O

DST = 176


def test_nearest_index_rule_vs_torch_and_exact_for_every_source_size():
    xs = np.arange(DST, dtype=np.int64)
    n_exact_points = n_oracle_below = n_torch_diff = checked = 0
    for src in range(1, 2001):
        got = O.resize_nearest_indices(DST, src)
        exact = np.minimum((xs * src) // DST, src - 1)
        tor = F.interpolate(torch.arange(src, dtype=torch.float32).reshape(1, 1, 1, src), size=(1, DST),
                            mode="nearest").reshape(-1).long().numpy()
        integer_pt = (xs * src) % DST == 0
        # away from exact-integer products the three implementations must agree index for index
        assert np.array_equal(got[~integer_pt], exact[~integer_pt]), src
        assert np.array_equal(got[~integer_pt], tor[~integer_pt]), src
        checked += int((~integer_pt).sum())
        # at exact-integer products: the double-reciprocal evaluation may fall one below, never elsewhere
        d = exact[integer_pt] - got[integer_pt]
        assert ((d == 0) | (d == 1)).all(), src
        n_exact_points += int(integer_pt.sum())
        n_oracle_below += int(d.sum())
        n_torch_diff += int((tor[integer_pt] != got[integer_pt]).sum())
        # and the oracle is the literal double evaluation of OpenCV's expression  cvFloor(x * (1. / fx)),
        # fx = (double)dst / src  (python floats are IEEE doubles; no numpy involved)
        ifx = 1.0 / (float(DST) / float(src))
        lit = [min(int(math.floor(x * ifx)), src - 1) for x in range(DST)]
        assert got.tolist() == lit, src
    assert checked > 330000
    # documented size of the evaluation-order-dependent set (informational)
    print("exact-integer points: %d of %d; oracle one below exact at %d; torch(float32 scale) differs at %d"
          % (n_exact_points, DST * 2000, n_oracle_below, n_torch_diff))
    assert 0 < n_oracle_below < n_exact_points


def test_nearest_rule_identity_and_integer_factors():
    # src == dst: identity; dst = k*src: every source pixel repeated k times; src = k*dst: every k-th pixel
    assert O.resize_nearest_indices(176, 176).tolist() == list(range(176))
    assert O.resize_nearest_indices(176, 88).tolist() == [i // 2 for i in range(176)]
    assert O.resize_nearest_indices(176, 352).tolist() == [2 * i for i in range(176)]
    assert O.resize_nearest_indices(176, 1).tolist() == [0] * 176
    img = np.arange(5 * 7 * 3).reshape(5, 7, 3).astype(np.uint8)
    out = O.resize_nearest(img, (176, 176))
    t = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(176, 176), mode="nearest")
    assert np.array_equal(out, t[0].permute(1, 2, 0).numpy().astype(np.uint8))


def _rotvecs():
    rng = np.random.default_rng(0)
    v = rng.normal(0, 1, (10000, 3)) * rng.uniform(0, 3.2, (10000, 1))
    v[:100] *= 1e-9                                                       # theta ~ 0 (scipy: Taylor branch)
    u = v[100:200] / np.linalg.norm(v[100:200], axis=1, keepdims=True)
    v[100:200] = u * (np.pi - rng.uniform(0, 1e-6, (100, 1)))             # theta ~ pi
    v[200] = 0.0                                                          # exactly zero -> identity
    v[201] = (1e-20, 0, 0)                                                # below DBL_EPSILON -> identity
    return v


def test_rodrigues_vs_scipy_rotation():
    e64 = e32 = 0.0
    for r in _rotvecs():
        R = O.rodrigues(r.astype(np.float64))
        S = Rotation.from_rotvec(r).as_matrix()
        assert R.dtype == np.float64
        e64 = max(e64, float(np.abs(R - S).max()))
        r32 = r.astype(np.float32)
        R32 = O.rodrigues(r32)                    # the reference's call: float32 vector -> float32 matrix
        assert R32.dtype == np.float32
        S32 = Rotation.from_rotvec(r32.astype(np.float64)).as_matrix()
        e32 = max(e32, float(np.abs(R32.astype(np.float64) - S32).max()))
        # a rotation: orthonormal, det +1
        assert abs(np.linalg.det(R) - 1.0) < 1e-12 and np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    assert e64 < 5e-15, e64            # double arithmetic, different algebraic route
    assert e32 < 6.0e-8, e32           # one float32 rounding of entries in [-1, 1] (2^-24 = 5.96e-8)
    assert np.array_equal(O.rodrigues(np.zeros(3, np.float32)), np.eye(3, dtype=np.float32))
    assert np.array_equal(O.rodrigues(np.array([1e-20, 0, 0], np.float32)), np.eye(3, dtype=np.float32))


def test_process_predict_vs_scipy_composition():
    """datasets.py:159-175 composed with scipy's rotation instead of the oracle's Rodrigues."""
    from oracle import fixtures as Fx
    rng = np.random.default_rng(5)
    for i in range(200):
        P = Fx.pose(70 + i)
        t = rng.uniform(-1, 1, 3).astype(np.float32)
        r = rng.uniform(-1, 1, 3).astype(np.float32)
        rn = (5 if i % 2 else 30) * np.pi / 180
        B = O.process_predict(P, t, r, 0.03, rn)
        rv = (r * np.float32(rn)).astype(np.float64)
        Bs = np.eye(4)
        Bs[:3, :3] = Rotation.from_rotvec(rv).as_matrix() @ P[:3, :3]
        Bs[:3, 3] = (t * np.float32(0.03)).astype(np.float64) + P[:3, 3]
        assert np.abs(B - Bs).max() < 1.2e-7       # the float32 cast of R (cv2.Rodrigues returns src depth)
