"""CPU: the Toom-Cook matrices compiled into wino_mfma.hip (parsed from the source) satisfy the
Winograd identity  Y = A^T [(G g G^T) (.) (B^T d B)] A  ==  valid 3x3 cross-correlation of the
(m+2)x(m+2) tile d with g, for m = 2, 4 and 6 -- and the float32 rounding of the transformed
algorithm stays in the class DESIGN.md quotes (rms error relative to the largest output)."""
import os
import re

import numpy as np
import pytest

SRC = os.path.join(os.path.dirname(__file__), "..", "iros20-6d-pose-tracking_amd", "csrc", "wino_mfma.hip")


def _matrix(src, func, name):
    body = src[src.index("constexpr " + ("double" if func == "wino_g" else "float") + " " + name, src.index(func + "(int i, int j)")):]
    body = body[body.index("=") + 1:body.index(";")]
    rows = re.findall(r"\{([^{}]+)\}", body)
    def num(tok):
        tok = tok.strip().rstrip("f")
        if "/" in tok:
            a, b = tok.split("/")
            return float(a) / float(b)
        return float(tok)
    return np.array([[num(t) for t in r.split(",")] for r in rows], dtype=np.float64)


@pytest.fixture(scope="module")
def mats():
    src = open(SRC).read()
    return {2: (_matrix(src, "wino_bt", "t2"), _matrix(src, "wino_g", "t2"), _matrix(src, "wino_at", "t2")),
            4: (_matrix(src, "wino_bt", "t4"), _matrix(src, "wino_g", "t4"), _matrix(src, "wino_at", "t4")),
            6: (_matrix(src, "wino_bt", "t6"), _matrix(src, "wino_g", "t6"), _matrix(src, "wino_at", "t6"))}


@pytest.mark.parametrize("m", [2, 4, 6])
def test_identity_in_float64(mats, m):
    BT, G, AT = mats[m]
    n = m + 2
    assert BT.shape == (n, n) and G.shape == (n, 3) and AT.shape == (m, n)
    rng = np.random.default_rng(m)
    for _ in range(20):
        d = rng.normal(size=(n, n))
        g = rng.normal(size=(3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        ref = np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(m)] for i in range(m)])
        assert np.abs(y - ref).max() < (1e-12 if m < 6 else 2e-11)   # the F(6x6) transforms amplify float64 rounding too


@pytest.mark.parametrize("m,bound", [(2, 5e-7), (4, 2e-6), (6, 5e-6)])
def test_float32_rounding_class(mats, m, bound):
    """One 512-channel layer in float32 (U, V rounded once, products accumulated in float32, transforms
    in float32) against float64: rms error / max |output|."""
    BT, G, AT = mats[m]
    n = m + 2
    rng = np.random.default_rng(10 + m)
    C, K, tiles = 512, 16, 9
    d = np.maximum(rng.normal(size=(tiles, C, n, n)), 0).astype(np.float32)          # post-ReLU like
    g = (rng.normal(size=(K, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    ref = np.zeros((tiles, K, m, m))
    for i in range(m):
        for j in range(m):
            ref[:, :, i, j] = np.einsum("tcrs,kcrs->tk", d[:, :, i:i + 3, j:j + 3].astype(np.float64), g.astype(np.float64))
    U = np.einsum("ir,kcrs,js->ijkc", G, g.astype(np.float64), G).astype(np.float32)
    f32 = np.float32
    V = np.einsum("ir,tcrs->tcis", BT.astype(f32), d).astype(f32)
    V = np.einsum("tcis,js->tcij", V, BT.astype(f32)).astype(f32)
    M = np.einsum("ijkc,tcij->tkij", U, V, dtype=f32)
    Y = np.einsum("pi,tkij->tkpj", AT.astype(f32), M).astype(f32)
    Y = np.einsum("tkpj,qj->tkpq", Y, AT.astype(f32)).astype(f32)
    rel_rms = float(np.sqrt(((Y - ref) ** 2).mean()) / np.abs(ref).max())
    assert rel_rms < bound, rel_rms
