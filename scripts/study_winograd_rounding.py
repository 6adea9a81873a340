#!/usr/bin/env python3
"""CPU study (no GPU): what would F(6x6,3x3) cost in float32 rounding?

The Winograd GEMMs of the 256 / 512-channel blocks lose 16 % to workgroup quantisation at batch 64 (3.375 tiles per slot,
profiles/EXPERIMENTS.md items 16); F(6x6,3x3) would give exactly 2.0 tiles per slot AND 21 % fewer multiplies and plane bytes.  Its
transforms amplify float32 rounding more than F(4x4)'s.  This script measures by how much, end to end, with the oracle network:
the four 256 / 512-channel stride-1 convs are replaced by a float32 emulation of the device algorithm (U = G g G^T in float64 rounded
once, B^T d B / the per-frequency products / A^T M A in float32) for m = 2, 4, 6 and compared with a float64 forward of the same
weights and inputs.  Output: max |d logit| per variant and the pose error it implies in the 30-degree regime (tolerance 1e-5).

    python scripts/study_winograd_rounding.py [pairs]
"""
import os
import sys
from fractions import Fraction

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from oracle import fixtures as Fx
from oracle import se3_oracle as O


def toom_cook(m, points):
    """A^T [m x n], G [n x 3], B^T [n x n] (n = m + 2) for F(m, 3) from n - 1 finite points + the point at infinity, in exact
    rationals (Toom-Cook / Lavin & Gray):  Y = A^T [(G g) (.) (B^T d)].
      A^T = transposed Vandermonde matrix of the points for polynomials of degree m - 1 (last column: infinity -> leading coefficient);
      G   = Vandermonde rows for degree 2, row i divided by N_i = prod_{k != i}(p_i - p_k)  (the Lagrange normalisation);
      B^T = row i: coefficients of prod_{k != i}(x - p_k);  last row: prod_k (x - p_k)."""
    n = m + 2
    pts = [Fraction(p) for p in points]
    assert len(pts) == n - 1

    def vander(cols):
        rows = [[p ** j for j in range(cols)] for p in pts]
        rows.append([Fraction(0)] * (cols - 1) + [Fraction(1)])
        return rows

    Am = vander(m)        # n x m
    AT = [[Am[i][j] for i in range(n)] for j in range(m)]
    G3 = vander(3)        # n x 3
    G = []
    for i in range(n - 1):
        Ni = Fraction(1)
        for k in range(n - 1):
            if k != i:
                Ni *= (pts[i] - pts[k])
        G.append([x / Ni for x in G3[i]])
    G.append(G3[n - 1])
    BT = []
    for i in range(n - 1):
        poly = [Fraction(1)]
        for k in range(n - 1):
            if k != i:
                poly = _polymul(poly, [-pts[k], Fraction(1)])
        BT.append(poly + [Fraction(0)] * (n - len(poly)))
    poly = [Fraction(1)]
    for k in range(n - 1):
        poly = _polymul(poly, [-pts[k], Fraction(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in r] for r in M], dtype=np.float64)
    return f(AT), f(G), f(BT)


def _polymul(a, b):
    out = [Fraction(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] += x * y
    return out


def check_identity(AT, G, BT, m):
    rng = np.random.default_rng(m)
    n = m + 2
    worst = 0.0
    for _ in range(10):
        d = rng.normal(size=(n, n)); g = rng.normal(size=(3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        ref = np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(m)] for i in range(m)])
        worst = max(worst, np.abs(y - ref).max() / np.abs(ref).max())
    return worst


POINTS = {2: [0, 1, -1], 4: [0, 1, -1, Fraction(1, 2), -2],
          6: [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2)],
          "6b": [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 2), Fraction(-3, 2)]}


class WinoConv:
    """float32 emulation of the device's Winograd conv for weight shapes [C, C, 3, 3], C in {256, 512}."""
    def __init__(self, key):
        m = int(str(key)[0])
        self.m = m
        AT, G, BT = toom_cook(m, POINTS[key])
        err = check_identity(AT, G, BT, m)
        assert err < 1e-9, err
        self.AT, self.G, self.BT = AT, G, BT
        self.cache = {}

    def __call__(self, x, w, b):
        m, n = self.m, self.m + 2
        N, C, H, W = x.shape
        th, tw = -(-H // m), -(-W // m)
        xp = F.pad(x, (1, tw * m + 1 - W, 1, th * m + 1 - H))
        d = xp.unfold(2, n, m).unfold(3, n, m)                     # [N, C, th, tw, n, n]
        BT = torch.from_numpy(self.BT).to(torch.float32); AT = torch.from_numpy(self.AT).to(torch.float32)
        V = torch.einsum("ir,nctwrs->nctwis", BT, d)
        V = torch.einsum("nctwis,js->nctwij", V, BT)
        key = w.data_ptr()
        if key not in self.cache:
            G = torch.from_numpy(self.G)
            self.cache[key] = torch.einsum("ir,kcrs,js->ijkc", G, w.double(), G).to(torch.float32)
        U = self.cache[key]
        M = torch.einsum("ijkc,nctwij->nktwij", U, V)
        Y = torch.einsum("pi,nktwij->nktwpj", AT, M)
        Y = torch.einsum("nktwpj,qj->nktwpq", Y, AT)
        Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(N, w.shape[0], th * m, tw * m)[:, :, :H, :W]
        return Y + b.view(1, -1, 1, 1)


def forward_with(sd, A, B, wino=None):
    real = F.conv2d
    def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
        if wino is not None and w.shape[-1] == 3 and stride == 1 and w.shape[0] == w.shape[1] and w.shape[0] in (256, 512):
            return wino(x, w, b)
        return real(x, w, b, stride, padding, *a, **k)
    F.conv2d = conv2d
    try:
        return O.forward(sd, A, B)
    finally:
        F.conv2d = real


def main():
    pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    rows = []
    for key in (2, 4, 6, "6b"):
        AT, G, BT = toom_cook(int(str(key)[0]), POINTS[key])
        rows.append((key, np.abs(BT).max(), np.abs(AT).max(), np.abs(G).max()))
    print("largest |entry| of B^T / A^T / G:", ", ".join("F(%sx%s) %.3g / %.3g / %.3g" % (str(k)[0], str(k)[0], b, a, g) for k, b, a, g in rows))
    worst = {}
    for seed, scale in ((1, 1.0), (5, 1.0), (11, 40.0)):
        sd = O.make_state_dict(seed % 4)
        A, B = Fx.net_inputs(seed, pairs, scale=scale) if scale != 1.0 else Fx.net_inputs(seed, pairs)
        sd64 = {k: v.double() for k, v in sd.items()}
        ref = O.forward(sd64, A.double(), B.double())
        ref_l = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
        variants = [("direct f32", None)] + [("F(%sx%s)%s" % (str(k)[0], str(k)[0], " alt points" if k == "6b" else ""), WinoConv(k)) for k in (2, 4, 6, "6b")]
        for name, w in variants:
            out = forward_with(sd, A, B, w)
            l = torch.cat([out["trans_logit"], out["rot_logit"]], 1).double()
            e = float((l - ref_l).abs().max())
            er = float((l[:, 3:] - ref_l[:, 3:]).abs().max())
            worst.setdefault(name, [0.0, 0.0])
            worst[name][0] = max(worst[name][0], e); worst[name][1] = max(worst[name][1], er)
            print("seed %2d x%-4g %-22s max |d logit| %.2e  (rot %.2e)   max |logit| %.2f" % (seed, scale, name, e, er, float(ref_l.abs().max())))
    print()
    base = worst["F(4x4)"][0]
    for name, (e, er) in worst.items():
        print("%-22s worst |d logit| %.2e = %.1f x F(4x4); implied |d pose| at 30 deg <= %.2e (rot logit error x 0.5236; tolerance 1e-5)"
              % (name, e, e / base, er * 0.5236))


if __name__ == "__main__":
    main()
