"""oracle/ss_rules.py evaluated array-at-a-time (TEST INFRASTRUCTURE ONLY).

ss_rules.py is the readable statement of the GL implementation's arithmetic -- one triangle at a time, confirmed against the live
library.  Rendering a 20,480-face mesh with it takes seconds; the closed-loop checkers (oracle/closed_loop.py, bench.py's `track`
block) render image A on the ORACLE side every frame, so this module evaluates the same statement over all triangles / pixels at
once with numpy float32 arrays (same operations, same order; float32 op float32 stays float32).  Triangles that cross the
frustum take ss_rules' own per-triangle path.  tests/test_gl_swiftshader.py holds both to the goldens' bytes and to each other."""
import numpy as np

from . import ss_rules as S

f32 = np.float32
_1 = f32(1)


def _setup_arrays(pv, faces, sub_bits):
    """per-triangle quantities of the unclipped triangles, vectorised.  Returns dict of arrays over ALL faces + masks."""
    F = np.asarray(faces, np.int64)
    X, Y, Z, w, rhw, flags = pv["X"], pv["Y"], pv["Z"], pv["w"], pv["rhw"], pv["flags"]
    i0, i1, i2 = F[:, 0], F[:, 1], F[:, 2]
    fl = np.stack([flags[i0], flags[i1], flags[i2]], 1)
    trivial = (fl[:, 0] & fl[:, 1] & fl[:, 2]) != 0
    needs_clip = (fl[:, 0] | fl[:, 1] | fl[:, 2]) != 0
    x = [X[i].astype(f32) for i in (i0, i1, i2)]
    y = [Y[i].astype(f32) for i in (i0, i1, i2)]
    A = ((y[2] - y[0]) * x[1] + (y[1] - y[2]) * x[0]) + (y[0] - y[1]) * x[2]
    neg = np.signbit(w[i0]) ^ np.signbit(w[i1]) ^ np.signbit(w[i2])
    A = np.where(neg, -A, A)
    d = A < 0
    Xi = np.stack([X[i0], X[i1], X[i2]], 1)
    Yi = np.stack([Y[i0], Y[i1], Y[i2]], 1)
    area2 = (Xi[:, 1] - Xi[:, 0]) * (Yi[:, 2] - Yi[:, 0]) - (Yi[:, 1] - Yi[:, 0]) * (Xi[:, 2] - Xi[:, 0])
    alive = ~trivial & (A != 0) & ~np.isnan(A)
    fast = alive & ~needs_clip & (area2 != 0) & ((area2 < 0) == d)
    slow = alive & needs_clip
    # rotation to the largest clip w: both conditions on the ORIGINAL order
    ww = np.stack([w[i0], w[i1], w[i2]], 1)
    wmax = np.maximum(np.maximum(ww[:, 0], ww[:, 1]), ww[:, 2])
    R = np.tile(np.array([0, 1, 2]), (len(F), 1))
    c1, c2 = wmax == ww[:, 1], wmax == ww[:, 2]
    R[c1] = R[c1][:, [1, 2, 0]]
    R[c2] = R[c2][:, [2, 0, 1]]
    rows = np.arange(len(F))
    vid = np.stack([F[rows, R[:, k]] for k in range(3)], 1)                 # rotated vertex ids
    sub = 1 << sub_bits
    X0, Y0 = X[vid[:, 0]], Y[vid[:, 0]]
    dx = X0.astype(f32) * f32(1.0 / sub)
    dy = Y0.astype(f32) * f32(1.0 / sub)
    X1, Y1 = X[vid[:, 1]] - X0, Y[vid[:, 1]] - Y0
    X2, Y2 = X[vid[:, 2]] - X0, Y[vid[:, 2]] - Y0
    fx1, fy1, fx2, fy2 = X1.astype(f32), Y1.astype(f32), X2.astype(f32), Y2.astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        z0 = Z[vid[:, 0]]
        z1, z2 = Z[vid[:, 1]] - z0, Z[vid[:, 2]] - z0
        D = _1 / (fx1 * fy2 - fx2 * fy1)
        zA = ((fy2 * z1 - fy1 * z2) * D) * f32(sub)
        zB = ((fx1 * z2 - fx2 * z1) * D) * f32(sub)
        zC = z0 * _1 + f32(0)
        w1, w2 = w[vid[:, 1]], w[vid[:, 2]]
        rhw0 = rhw[vid[:, 0]]
        sc = f32(1.0 / sub)
        px1, py1 = (w1 * sc) * fx1, (w1 * sc) * fy1
        px2, py2 = (w2 * sc) * fx2, (w2 * sc) * fy2
        a = px1 * py2 - px2 * py1
        Ai = _1 / a
        Dm = Ai * rhw0
        M = np.zeros((len(F), 3, 3), f32)
        M[:, 0, 2] = rhw0
        nz = a != 0
        M[:, 0, 0] = np.where(nz, (py1 * w2 - py2 * w1) * Dm, f32(0))
        M[:, 0, 1] = np.where(nz, (px2 * w1 - px1 * w2) * Dm, f32(0))
        M[:, 1, 0] = np.where(nz, py2 * Ai, f32(0))
        M[:, 1, 1] = np.where(nz, -px2 * Ai, f32(0))
        M[:, 2, 0] = np.where(nz, -py1 * Ai, f32(0))
        M[:, 2, 1] = np.where(nz, px1 * Ai, f32(0))
    return dict(fast=fast, slow=slow, Xi=Xi, Yi=Yi, area2=area2, vid=vid, dx=dx, dy=dy, zA=zA, zB=zB, zC=zC, M=M)


def _quad(v, d):
    return (v // 2 * 2).astype(f32) + ((v % 2).astype(f32) - d)


def rasterize(pv, faces, W, H):
    """-> (zbuf float32 [H,W] (1.0 = cleared), owner int32 [H,W] (-1 = none), per-triangle arrays)"""
    sb = pv.get("sub_bits", 4)
    T = _setup_arrays(pv, faces, sb)
    idx = np.nonzero(T["fast"])[0]
    Xi, Yi = T["Xi"][idx].copy(), T["Yi"][idx].copy()
    flip = T["area2"][idx] < 0
    Xi[flip] = Xi[flip][:, [0, 2, 1]]
    Yi[flip] = Yi[flip][:, [0, 2, 1]]
    sm = (1 << sb) - 1
    x0 = np.maximum((Xi.min(1) + sm) >> sb, 0); x1 = np.minimum((Xi.max(1) + sm) >> sb, W)
    y0 = np.maximum((Yi.min(1) + sm) >> sb, 0); y1 = np.minimum((Yi.max(1) + sm) >> sb, H)
    bw, bh = np.maximum(x1 - x0, 0), np.maximum(y1 - y0, 0)
    cnt = bw * bh
    keep = cnt > 0
    idx, Xi, Yi, x0, y0, bw, cnt = idx[keep], Xi[keep], Yi[keep], x0[keep], y0[keep], bw[keep], cnt[keep]
    tri = np.repeat(np.arange(len(idx)), cnt)                          # candidate -> position in idx
    k = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)    # running index inside the bounding box
    xs = x0[tri] + k % bw[tri]
    ys = y0[tri] + k // bw[tri]
    px, py = xs << sb, ys << sb
    inside = np.ones(len(tri), bool)
    for a_, b_ in ((0, 1), (1, 2), (2, 0)):
        ddx = (Xi[:, b_] - Xi[:, a_])[tri]; ddy = (Yi[:, b_] - Yi[:, a_])[tri]
        E = ddx * (py - Yi[tri, a_]) - ddy * (px - Xi[tri, a_])
        inside &= (E > 0) | ((E == 0) & ((ddy < 0) | ((ddy == 0) & (ddx > 0))))
    tri, xs, ys = tri[inside], xs[inside], ys[inside]
    t = idx[tri]
    z = (T["zC"][t] + _quad(ys, T["dy"][t]) * T["zB"][t]) + _quad(xs, T["dx"][t]) * T["zA"][t]
    cand_t, cand_x, cand_y, cand_z = [t], [xs], [ys], [z]
    slow_setups = {}
    for ts in np.nonzero(T["slow"])[0]:                                 # triangles that cross the frustum: ss_rules' own path
        s = S.setup_triangle(pv, np.asarray(faces)[ts], W, H)
        if not s.ok:
            continue
        ymin, ymax = s.rows
        l, r = s.x0[ymin:ymax], s.x1[ymin:ymax]
        c = np.maximum(r - l, 0)
        if c.sum() == 0:
            continue
        yy = np.repeat(np.arange(ymin, ymax), c)
        xx = np.concatenate([np.arange(a, b) for a, b in zip(l, r) if b > a])
        qx, qy = S._quad_coords(xx, yy, s.dx, s.dy)
        cand_t.append(np.full(len(xx), ts)); cand_x.append(xx); cand_y.append(yy)
        cand_z.append(S._plane_eval((s.zA, s.zB, s.zC), qx, qy))
        slow_setups[int(ts)] = s
    t, xs, ys, z = np.concatenate(cand_t), np.concatenate(cand_x), np.concatenate(cand_y), np.concatenate(cand_z)
    ok = z < _1                                                         # GL_LESS against the cleared 1.0 (NaN fails)
    t, xs, ys, z = t[ok], xs[ok], ys[ok], z[ok]
    pix = ys * W + xs
    order = np.lexsort((t, z, pix))                                     # per pixel: smallest z, then the earliest triangle
    pix, t, z = pix[order], t[order], z[order]
    first = np.r_[True, pix[1:] != pix[:-1]] if len(pix) else np.zeros(0, bool)
    zbuf = np.full(H * W, _1, f32)
    owner = np.full(H * W, -1, np.int32)
    zbuf[pix[first]] = z[first]
    owner[pix[first]] = t[first]
    return zbuf.reshape(H, W), owner.reshape(H, W), T


def interpolate(T, owner, attrs):
    """perspective-correct varyings at every covered pixel: attrs [V, C] float32 -> (ys, xs, values [n, C])"""
    ys, xs = np.nonzero(owner >= 0)
    t = owner[ys, xs]
    M = T["M"][t]
    xx, yy = _quad(xs, T["dx"][t]), _quad(ys, T["dy"][t])
    vid = T["vid"][t]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        Pw = (M[:, 0] + M[:, 1]) + M[:, 2]
        wv = (Pw[:, 2] + yy * Pw[:, 1]) + xx * Pw[:, 0]
        rcp = _1 / wv
        rcp = (rcp + rcp) - (wv * rcp) * rcp
        out = np.zeros((len(xs), attrs.shape[1]), f32)
        a0, a1, a2 = attrs[vid[:, 0]], attrs[vid[:, 1]], attrs[vid[:, 2]]
        for c in range(attrs.shape[1]):
            P = (a0[:, c:c + 1] * M[:, 0] + a1[:, c:c + 1] * M[:, 1]) + a2[:, c:c + 1] * M[:, 2]
            out[:, c] = ((P[:, 2] + yy * P[:, 1]) + xx * P[:, 0]) * rcp
    return ys, xs, out


def render_vispy(vertices, normals, colors01, faces, ob2cam, K, window, size=176, numpy_rule="numpy2", sub_bits=4):
    """ss_rules.render_vispy, vectorised: rgb uint8 [size,size,3], depth uint16 [size,size]"""
    W = H = size
    P, V, light, (pA, pB) = S.vispy_uniforms(ob2cam, K, window)
    v32 = np.asarray(vertices, f32)
    pv = S.project(S.clip_positions(v32, S._mat_mul_cols(P, V)), W, H, sub_bits)
    zbuf, owner, T = rasterize(pv, faces, W, H)
    if T["slow"].any():
        # (a clipped triangle's planes are those of the unclipped triangle: the arrays hold them already)
        pass
    attrs = np.concatenate([v32, np.asarray(normals, f32), np.asarray(colors01, f32)], 1)
    ys, xs, a = interpolate(T, owner, attrs)
    colf = np.zeros((H, W, 3), f32)
    colf[ys, xs] = S.shade_vispy(a[:, 0:3], a[:, 3:6], a[:, 6:9], light)
    rgb = S.unorm8(colf)
    rgb[owner < 0] = 0
    return rgb, S.depth_mm(zbuf, pA, pB, numpy_rule)
