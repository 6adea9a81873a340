"""oracle/ss_rules.py (the operation-by-operation statement of the software GL behind the renderer goldens) against the LIVE
library, bit for bit, on random single triangles drawn through a pass-through program into float colour / float depth
attachments (oracle/ss_probe.py).  Needs the kaleido wheel's SwiftShader (present in this image); skipped elsewhere -- the
golden-based tests (tests/test_gl_swiftshader.py) do not need it.  These are the experiments each rule was fitted with."""
import numpy as np
import pytest

from oracle import ss_rules as S
from oracle import swiftshader_gl as SG

pytestmark = pytest.mark.skipif(not SG.available(), reason="needs the kaleido wheel's SwiftShader")
f32 = np.float32
W = H = 32
TRI = np.array([[0, 1, 2]])


@pytest.fixture(scope="module")
def probe():
    from oracle.ss_probe import Probe
    return Probe(W, H)


def _clip(rng, lo, hi, wlo=0.3, whi=2.0, zlo=-0.9, zhi=0.9):
    tri = rng.uniform(lo, hi, (3, 2))
    w = rng.uniform(wlo, whi, 3)
    zn = rng.uniform(zlo, zhi, 3)
    return np.c_[(2 * tri[:, 0] / W - 1) * w, (2 * tri[:, 1] / H - 1) * w, zn * w, w].astype(f32)


def _check(probe, pos, attr=None):
    col, z = probe.draw(pos, attr)
    zb, ow, st = S.rasterize(S.project(pos, W, H), TRI, W, H)
    assert np.array_equal(ow >= 0, z < 1), "coverage"
    assert np.array_equal(zb.view(np.int32), z.view(np.int32)), "depth bits"
    if attr is not None and 0 in st:
        ys, xs = np.nonzero(ow == 0)
        got = S.interpolate(st[0], attr[st[0].idx], xs, ys)
        assert np.array_equal(got.view(np.int32), col[ys, xs].view(np.int32)), "varying bits"
    return int((ow >= 0).sum())


def test_coverage_depth_and_varyings_inside_the_viewport(probe):
    rng = np.random.default_rng(0)
    px = 0
    for _ in range(150):
        px += _check(probe, _clip(rng, 1, W - 1), rng.uniform(-1, 1, (3, 4)).astype(f32))
    assert px > 5000


def test_fill_rule_on_exact_ties(probe):
    """vertices on the half-pixel grid: pixel centres ON edges and vertices; the top-left rule decides"""
    rng = np.random.default_rng(1)
    for _ in range(150):
        tri = rng.integers(0, 2 * W + 1, (3, 2)) / 2.0
        _check(probe, np.c_[2 * tri[:, 0] / W - 1, 2 * tri[:, 1] / H - 1, np.zeros(3), np.ones(3)].astype(f32))


def test_equal_w_rotation_rule(probe):
    """which vertex the plane equations are anchored at when clip w ties (both conditions use the ORIGINAL order)"""
    rng = np.random.default_rng(2)
    for it in range(120):
        pos = _clip(rng, 1, W - 1)
        w = pos[:, 3].copy()
        k = it % 4
        if k == 0:
            w[1] = w[2] = w.max()
        elif k == 1:
            w[0] = w[1] = w.max()
        elif k == 2:
            w[0] = w[2] = w.max()
        else:
            w[:] = w[0]
        pos = (pos / pos[:, 3:4] * w[:, None]).astype(f32)
        _check(probe, pos, rng.uniform(-1, 1, (3, 4)).astype(f32))


@pytest.mark.parametrize("what", ["xy", "z", "xyz", "negative_w"])
def test_clipping(probe, what):
    rng = np.random.default_rng(3)
    drawn = 0
    for _ in range(150):
        if what == "xy":
            pos = _clip(rng, -W, 2 * W)
        elif what == "z":
            pos = _clip(rng, 1, W - 1, zlo=-1.5, zhi=1.5)
        elif what == "xyz":
            pos = _clip(rng, -W, 2 * W, zlo=-1.5, zhi=1.5)
        else:
            pos = _clip(rng, -W, 2 * W, wlo=-1.0, zlo=-1.5, zhi=1.5)
        drawn += _check(probe, pos) > 0
    assert drawn > 30


def test_unorm8_conversion(probe):
    """float colour -> the byte an RGBA8 target stores"""
    from oracle.ss_probe import Probe
    p8 = Probe(W, H, color_float=False)
    rng = np.random.default_rng(4)
    n = 0
    for _ in range(40):
        pos = _clip(rng, 1, W - 1)
        attr = rng.uniform(-0.1, 1.1, (3, 4)).astype(f32)
        colf, z = probe.draw(pos, attr)
        col8, z8 = p8.draw(pos, attr)
        m = z < 1
        assert np.array_equal(z, z8)
        assert np.array_equal(S.unorm8(colf[m]), col8[m])
        n += int(m.sum())
    assert n > 1500


def test_depth_test_is_less_in_draw_order(probe):
    """two coplanar copies of one triangle with different varyings: the FIRST one drawn owns every pixel"""
    rng = np.random.default_rng(5)
    pos = _clip(rng, 2, W - 2)
    pos2 = np.concatenate([pos, pos])
    attr = np.concatenate([np.full((3, 4), 0.25, f32), np.full((3, 4), 0.75, f32)])
    col, z = probe.draw(pos2, attr, faces=np.array([[0, 1, 2], [3, 4, 5]]))
    m = z < 1
    assert m.sum() > 20 and np.allclose(col[m], 0.25, atol=1e-6)
    zb, ow, _ = S.rasterize(S.project(pos2, W, H), np.array([[0, 1, 2], [3, 4, 5]]), W, H)
    assert (ow[m] == 0).all()
