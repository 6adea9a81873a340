"""Arithmetic model of the software OpenGL the renderer goldens were made on (TEST INFRASTRUCTURE ONLY).

tests/golden/gl_swiftshader.npz is what the UNMODIFIED reference class VispyRenderer (vispy_renderer.py:47-178) renders on
Google SwiftShader 4.1.0.7 (OpenGL ES 3.0).  OpenGL leaves rasterisation details implementation-defined (sub-pixel precision,
interpolation arithmetic, float -> unorm conversion), so "the reference's bytes" for image A are that implementation's bytes.
This module states that implementation's fixed-function arithmetic operation by operation in float32 / integers -- every rule
below was fitted as a hypothesis and then CONFIRMED BIT FOR BIT against the live library with oracle/ss_probe.py
(tests/test_ss_rules.py repeats the confirmation wherever the library is present):

  vertex      clip = (P V) p with P V formed per column, products accumulated left to right, no fused multiply-add;
              depth clip z' = (z + w) / 2;  rhw = 1 / w;
              X = rint(X0x16 + (x rhw) Wx16), Y likewise, Wx16 = 8 W, X0x16 = 8 W - 8: window coordinates in 1/16 pixel
              (GL_SUBPIXEL_BITS = 4) with the pixel centres on multiples of 16; round half to even;  Z = z' rhw
  coverage    exact integers on (X, Y): rows ceil(Ymin / 16) <= y < ceil(Ymax / 16), columns ceil(xl(y)) <= x < ceil(xr(y))
              == edge functions with E > 0, or E == 0 on edges with dy < 0 or (dy == 0 and dx > 0) (orientation made positive)
  rotation    the triangle's vertices are rotated so that v0 has the largest clip w (w1 == wmax: rotate left; ORIGINAL
              w2 == wmax: rotate right -- with w1 == w2 == wmax the two cancel)
  depth       z = (C + (y - Y0/16) B) + (x - X0/16) A with A = 16 (y2 z1 - y1 z2) D, B = 16 (x1 z2 - x2 z1) D, D = 1 / (x1 y2 - x2 y1)
              on the integer deltas converted to float; evaluated per 2 x 2 quad as (float(xq) + (xoff - X0/16)); LESS, in draw order
  varyings    plane equations of attribute / w from M (below), w interpolated as 1 / w, rcp = 1 / w then one Newton step
              (rcp + rcp) - (w rcp) rcp, value = plane(x, y) * rcp
  clipping    Sutherland-Hodgman in clip space, planes in the order near, far, left, right, top, bottom;
              new vertex = (dj Vi - di Vj) / (dj - di) with the clipped coordinate then set exactly; the polygon's vertices are
              snapped like triangle vertices and its outline walked edge by edge; plane equations stay those of the triangle
  colour      float -> UNORM8 through 16 bits: c16 = trunc(clamp(c) 65535), c8 = (c16 - (c16 >> 8) + 128) >> 8

`render_vispy` runs the reference's shader pair (vispy_renderer.py:54-98) through that model; the product's HIP rasteriser
(csrc/raster.hip, rule SE3TN_RASTER_RULE_SWIFTSHADER) implements the same statement and is compared with both."""
import numpy as np

f32 = np.float32
NEAR, FAR = 0.1, 2.0


def _mat_mul_cols(P, V):
    """GLSL mat4 * mat4 as the shader executes it: result column j = sum_k P[:, k] V[k, j], accumulated left to right."""
    R = np.zeros((4, 4), f32)
    for j in range(4):
        acc = (P[:, 0] * V[0, j]).astype(f32)
        for k in range(1, 4):
            acc = (acc + (P[:, k] * V[k, j]).astype(f32)).astype(f32)
        R[:, j] = acc
    return R


def clip_positions(verts, PV):
    """gl_Position = PV * vec4(p, 1): columns scaled by the components, accumulated left to right.  verts [n,3] f32 -> [n,4] f32"""
    v = np.asarray(verts, f32)
    acc = (PV[None, :, 0] * v[:, 0:1]).astype(f32)
    acc = (acc + (PV[None, :, 1] * v[:, 1:2]).astype(f32)).astype(f32)
    acc = (acc + (PV[None, :, 2] * v[:, 2:3]).astype(f32)).astype(f32)
    acc = (acc + (PV[None, :, 3] * f32(1)).astype(f32)).astype(f32)
    return acc


def project(clip, W, H, sub_bits=4):
    """post-transform position, snapped window coordinates (1 / 2^sub_bits pixel), depth and 1/w per vertex"""
    sub = 1 << sub_bits
    x, y, z, w = [np.ascontiguousarray(clip[:, i], f32) for i in range(4)]
    zc = ((z + w).astype(f32) * f32(0.5)).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        wsafe = np.where(w == 0, f32(1), w).astype(f32)
        rhw = (f32(1) / wsafe).astype(f32)
        Wx16, Hx16 = f32(W * 0.5 * sub), f32(H * 0.5 * sub)
        X0x16, Y0x16 = f32(W * 0.5 * sub - sub / 2), f32(H * 0.5 * sub - sub / 2)
        Xf = (X0x16 + ((x * rhw).astype(f32) * Wx16).astype(f32)).astype(f32)
        Yf = (Y0x16 + ((y * rhw).astype(f32) * Hx16).astype(f32)).astype(f32)
        X = np.where(np.abs(Xf) < 2.0 ** 30, np.rint(Xf), -2.0 ** 31).astype(np.int64)
        Y = np.where(np.abs(Yf) < 2.0 ** 30, np.rint(Yf), -2.0 ** 31).astype(np.int64)
        Z = (zc * rhw).astype(f32)
    flags = ((x > w) * 1) | ((y > w) * 2) | ((zc > w) * 4) | ((x < -w) * 8) | ((y < -w) * 16) | ((zc < 0) * 32)
    post = np.stack([x, y, zc, w], 1)
    return dict(post=post, X=X, Y=Y, Z=Z, rhw=rhw, w=w, flags=flags.astype(np.int32), sub_bits=sub_bits)


CLIP_RIGHT, CLIP_TOP, CLIP_FAR, CLIP_LEFT, CLIP_BOTTOM, CLIP_NEAR = 1, 2, 4, 8, 16, 32


def _clip_edge(Vi, Vj, di, dj):
    D = f32(f32(1) / f32(dj - di))
    return (((dj * Vi).astype(f32) - (di * Vj).astype(f32)).astype(f32) * D).astype(f32)


def clip_polygon(poly, flags_or):
    """poly: list of float32[4] post-transform positions.  Returns the clipped polygon (list), possibly with < 3 vertices."""
    planes = [(CLIP_NEAR, lambda v: v[2], 2, lambda b: f32(0)),
              (CLIP_FAR, lambda v: f32(v[3] - v[2]), 2, lambda b: b[3]),
              (CLIP_LEFT, lambda v: f32(v[3] + v[0]), 0, lambda b: f32(-b[3])),
              (CLIP_RIGHT, lambda v: f32(v[3] - v[0]), 0, lambda b: b[3]),
              (CLIP_TOP, lambda v: f32(v[3] - v[1]), 1, lambda b: b[3]),
              (CLIP_BOTTOM, lambda v: f32(v[3] + v[1]), 1, lambda b: f32(-b[3]))]
    for flag, dist, comp, exact in planes:
        if not (flags_or & flag):
            continue
        if len(poly) < 3:
            break
        out = []
        n = len(poly)
        for i in range(n):
            j = 0 if i == n - 1 else i + 1
            di, dj = f32(dist(poly[i])), f32(dist(poly[j]))
            if di >= 0:
                out.append(poly[i])
                if dj < 0:
                    b = _clip_edge(poly[i], poly[j], di, dj)
                    b[comp] = exact(b)
                    out.append(b)
            elif dj > 0:
                b = _clip_edge(poly[j], poly[i], dj, di)
                b[comp] = exact(b)
                out.append(b)
        poly = out
    return poly


def _snap_polygon(poly, W, H, sub_bits=4):
    Xs, Ys = [], []
    sub = 1 << sub_bits
    Wx16, Hx16 = f32(W * 0.5 * sub), f32(H * 0.5 * sub)
    X0x16, Y0x16 = f32(W * 0.5 * sub - sub / 2), f32(H * 0.5 * sub - sub / 2)
    for v in poly:
        w = f32(v[3])
        rhw = f32(1) / w if w != 0 else f32(1)
        Xs.append(int(np.rint(f32(X0x16 + f32(f32(v[0] * rhw) * Wx16)))))
        Ys.append(int(np.rint(f32(Y0x16 + f32(f32(v[1] * rhw) * Hx16)))))
    return Xs, Ys


def _cdiv(a, b):            # C division: truncation towards zero
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b > 0) else -q


def outline(Xs, Ys, d, W, H, sub_bits=4):
    """The span tables an n-gon's edges leave behind, walked in the implementation's order.  d = 1 if the snapped TRIANGLE's
    area (y2-y0) x1 + (y1-y2) x0 + (y0-y1) x2 (sign-corrected by the w signs) is < 0, else 0.  Returns (left, right, y0, y1)."""
    n = len(Xs)
    sb, sm = sub_bits, (1 << sub_bits) - 1
    left = np.zeros(H + 1, np.int64)
    right = np.zeros(H + 1, np.int64)
    Xq, Yq = list(Xs) + [Xs[0]], list(Ys) + [Ys[0]]
    for i in range(n):
        Xa, Ya, Xb, Yb = Xq[i + 1 - d], Yq[i + 1 - d], Xq[i + d], Yq[i + d]
        if Ya == Yb:
            continue
        swap = Yb < Ya
        X1, Y1, X2, Y2 = (Xb, Yb, Xa, Ya) if swap else (Xa, Ya, Xb, Yb)
        y1 = max((Y1 + sm) >> sb, 0)
        y2 = min((Y2 + sm) >> sb, H)
        if y1 >= y2:
            continue
        table = right if swap else left
        DX, DY = X2 - X1, Y2 - Y1
        FDX, FDY = DX << sb, DY << sb
        Xn = DX * ((y1 << sb) - Y1) + (X1 & sm) * DY
        x = (X1 >> sb) + _cdiv(Xn, FDY)
        dd = Xn - _cdiv(Xn, FDY) * FDY
        if dd > 0:
            x += 1
            dd -= FDY
        Q = _cdiv(FDX, FDY)
        R = FDX - Q * FDY
        if R < 0:
            Q -= 1
            R += FDY
        for y in range(y1, y2):
            table[y] = min(max(x, 0), W)
            x += Q
            dd += R
            if dd > 0:
                dd -= FDY
                x += 1
    ymin = max((min(Ys) + sm) >> sb, 0)
    ymax = min((max(Ys) + sm) >> sb, H)
    return left, right, ymin, ymax


def _rotate_max_w(idx, w):
    """v0 <- the vertex with the largest clip w.  Both conditions are evaluated on the ORIGINAL order (w1 == wmax rotates left,
    then w2 == wmax rotates right): with w1 == w2 == wmax the two rotations cancel and v0 stays."""
    o = list(idx)
    i = list(idx)
    wmax = max(w[o[0]], w[o[1]], w[o[2]])
    if wmax == w[o[1]]:
        i = [i[1], i[2], i[0]]
    if wmax == w[o[2]]:
        i = [i[2], i[0], i[1]]
    return i


class Setup:
    """Everything the implementation derives once per triangle."""
    __slots__ = ("ok", "rows", "x0", "x1", "dx", "dy", "zA", "zB", "zC", "M", "idx")


def setup_triangle(pv, tri, W, H):
    X, Y, Z, w, rhw, flags, post = pv["X"], pv["Y"], pv["Z"], pv["w"], pv["rhw"], pv["flags"], pv["post"]
    s = Setup()
    s.ok = False
    sb = pv.get("sub_bits", 4)
    sub = 1 << sb
    i0, i1, i2 = [int(t) for t in tri]
    if flags[i0] & flags[i1] & flags[i2]:
        return s
    x0, x1, x2 = f32(X[i0]), f32(X[i1]), f32(X[i2])
    y0, y1, y2 = f32(Y[i0]), f32(Y[i1]), f32(Y[i2])
    A = f32(f32(f32(f32(y2 - y0) * x1) + f32(f32(y1 - y2) * x0)) + f32(f32(y0 - y1) * x2))
    if A == 0:
        return s
    neg = (w[i0] < 0) ^ (w[i1] < 0) ^ (w[i2] < 0)
    if neg:
        A = -A
    d = 1 if A < 0 else 0
    flags_or = int(flags[i0] | flags[i1] | flags[i2])
    if flags_or:
        poly = clip_polygon([post[i0].copy(), post[i1].copy(), post[i2].copy()], flags_or)
        if len(poly) < 3:
            return s
        Xs, Ys = _snap_polygon(poly, W, H, sb)
    else:
        Xs, Ys = [int(X[i0]), int(X[i1]), int(X[i2])], [int(Y[i0]), int(Y[i1]), int(Y[i2])]
    left, right, ymin, ymax = outline(Xs, Ys, d, W, H, sb)
    while ymin < ymax and left[ymin] == right[ymin]:
        ymin += 1
    while ymax > ymin and left[ymax - 1] == right[ymax - 1]:
        ymax -= 1
    if ymin == ymax:
        return s
    s.rows = (ymin, ymax)
    s.x0, s.x1 = left, right
    r = _rotate_max_w([i0, i1, i2], w)
    s.idx = r
    X0, X1, X2 = [int(X[k]) for k in r]
    Y0, Y1, Y2 = [int(Y[k]) for k in r]
    w0, w1, w2 = [f32(w[k]) for k in r]
    rhw0 = f32(rhw[r[0]])
    s.dx, s.dy = f32(f32(X0) * f32(1.0 / sub)), f32(f32(Y0) * f32(1.0 / sub))
    X1 -= X0; Y1 -= Y0; X2 -= X0; Y2 -= Y0
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        sc = f32(1.0 / sub)
        px1, py1 = f32(f32(w1 * sc) * f32(X1)), f32(f32(w1 * sc) * f32(Y1))
        px2, py2 = f32(f32(w2 * sc) * f32(X2)), f32(f32(w2 * sc) * f32(Y2))
        a = f32(f32(px1 * py2) - f32(px2 * py1))
        M = np.zeros((3, 3), f32)
        M[0, 2] = rhw0
        if a != 0:
            Ai = f32(f32(1) / a)
            D = f32(Ai * rhw0)
            M[0, 0] = f32(f32(f32(py1 * w2) - f32(py2 * w1)) * D)
            M[0, 1] = f32(f32(f32(px2 * w1) - f32(px1 * w2)) * D)
            M[1, 0] = f32(py2 * Ai)
            M[1, 1] = f32(f32(-px2) * Ai)
            M[2, 0] = f32(f32(-py1) * Ai)
            M[2, 1] = f32(px1 * Ai)
        s.M = M
        z0, z1, z2 = [f32(Z[k]) for k in r]
        z1, z2 = f32(z1 - z0), f32(z2 - z0)
        fx1, fy1, fx2, fy2 = f32(X1), f32(Y1), f32(X2), f32(Y2)
        D = f32(f32(1) / f32(f32(fx1 * fy2) - f32(fx2 * fy1)))
        s.zA = f32(f32(f32(f32(fy2 * z1) - f32(fy1 * z2)) * D) * f32(sub))
        s.zB = f32(f32(f32(f32(fx1 * z2) - f32(fx2 * z1)) * D) * f32(sub))
        s.zC = f32(f32(z0 * f32(1)) + f32(0))
    s.ok = True
    return s


def _quad_coords(xs, ys, dx, dy):
    """(x - X0/16), (y - Y0/16) the way the 2 x 2 quad loop forms them: float(even) + (odd - d)"""
    xq, xo = (xs // 2 * 2).astype(f32), (xs % 2).astype(f32)
    yq, yo = (ys // 2 * 2).astype(f32), (ys % 2).astype(f32)
    return (xq + (xo - dx).astype(f32)).astype(f32), (yq + (yo - dy).astype(f32)).astype(f32)


def _plane_eval(P, xx, yy):
    return ((P[2] + (yy * P[1]).astype(f32)).astype(f32) + (xx * P[0]).astype(f32)).astype(f32)


def rasterize(pv, faces, W, H, depth_clamp_test=True):
    """z-buffer pass: returns (zbuf float32 [H,W] with 1.0 = cleared, owner int32 [H,W] (-1 = none), setups)"""
    zbuf = np.full((H, W), f32(1.0), f32)
    owner = np.full((H, W), -1, np.int32)
    setups = {}
    for t, tri in enumerate(np.asarray(faces)):
        s = setup_triangle(pv, tri, W, H)
        if not s.ok:
            continue
        ymin, ymax = s.rows
        rows = np.arange(ymin, ymax)
        l, r = s.x0[ymin:ymax], s.x1[ymin:ymax]
        cnt = np.maximum(r - l, 0)
        if cnt.sum() == 0:
            continue
        ys = np.repeat(rows, cnt)
        xs = np.concatenate([np.arange(a, b) for a, b in zip(l, r) if b > a])
        xx, yy = _quad_coords(xs, ys, s.dx, s.dy)
        z = _plane_eval((s.zA, s.zB, s.zC), xx, yy)
        win = z < zbuf[ys, xs]                                  # GL_LESS against the stored float (cleared 1.0)
        if not win.any():
            continue
        ys, xs, z = ys[win], xs[win], z[win]
        zbuf[ys, xs] = z
        owner[ys, xs] = t
        setups[t] = s
    return zbuf, owner, setups


def interpolate(s, attrs, xs, ys):
    """perspective-correct varyings of triangle setup `s` at pixels (xs, ys); attrs [3 (triangle order), C] float32"""
    M = s.M
    xx, yy = _quad_coords(xs, ys, s.dx, s.dy)
    Pw = ((M[0] + M[1]).astype(f32) + M[2]).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        wv = _plane_eval(Pw, xx, yy)
        rcp = (f32(1) / wv).astype(f32)
        rcp = ((rcp + rcp).astype(f32) - ((wv * rcp).astype(f32) * rcp).astype(f32)).astype(f32)
        out = np.zeros((len(xs), attrs.shape[1]), f32)
        for c in range(attrs.shape[1]):
            P = (((attrs[0, c] * M[0]).astype(f32) + (attrs[1, c] * M[1]).astype(f32)).astype(f32) + (attrs[2, c] * M[2]).astype(f32)).astype(f32)
            out[:, c] = (_plane_eval(P, xx, yy) * rcp).astype(f32)
    return out


def unorm8(c):
    """float colour -> the byte an 8-bit unorm target stores: through a 16-bit fixed-point value"""
    c = np.minimum(np.maximum(np.asarray(c, f32), f32(0)), f32(1))
    c16 = np.trunc((c * f32(65535.0)).astype(f32)).astype(np.int64)
    return ((c16 - (c16 >> 8) + 128) >> 8).astype(np.uint8)


def vispy_uniforms(ob2cam, K, window):
    """The float32 uniforms the reference uploads: proj (update_cam_mat, vispy_renderer.py:135-150), view and light_direction
    (render_image :171-176, with ob2cam_gl = inv(glcam_in_cvcam) . ob2cam, predict.py:205-207).  Returned as the GLSL matrices."""
    left, top, right, bottom = [float(v) for v in window]
    n, f = NEAR, FAR
    proj = np.array([[K[0, 0], 0, -K[0, 2], 0], [0, K[1, 1], -K[1, 2], 0], [0, 0, n + f, n * f], [0, 0, -1, 0]])
    ortho = np.array([[2. / (right - left), 0, 0, -(right + left) / (right - left)],
                      [0, 2. / (top - bottom), 0, -(top + bottom) / (top - bottom)],
                      [0, 0, -2 / (f - n), -(f + n) / (f - n)], [0, 0, 0, 1]]).astype(np.float32)
    pm = ortho.dot(proj).T                                       # what the reference stores; uploaded with transpose = FALSE
    glcam_in_cvcam = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
    ob2cam_gl = np.linalg.inv(glcam_in_cvcam).dot(np.asarray(ob2cam, np.float64))
    light = np.dot(np.linalg.inv(ob2cam_gl.T), np.array([0, 0.1, -0.9, 1]))[:3].astype(np.float32)
    P = np.ascontiguousarray(pm, np.float32).T                    # GLSL matrix = transpose of the uploaded row-major bytes
    V = np.ascontiguousarray(ob2cam_gl.T, np.float32).T
    return P, V, light, (pm[2, 2], pm[3, 2])


def shade_vispy(pos, nrm, col, light):
    """the reference's fragment shader (vispy_renderer.py:54-76) in float32, operation by operation"""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        nl = (-light).astype(f32)
        d = (nl[None, :] - pos).astype(f32)
        dot = (((d[:, 0] * d[:, 0]).astype(f32) + (d[:, 1] * d[:, 1]).astype(f32)).astype(f32) + (d[:, 2] * d[:, 2]).astype(f32)).astype(f32)
        rsq = (f32(1) / np.sqrt(dot).astype(f32)).astype(f32)
        ld = (d * rsq[:, None]).astype(f32)
        ndl = (((nrm[:, 0] * ld[:, 0]).astype(f32) + (nrm[:, 1] * ld[:, 1]).astype(f32)).astype(f32) + (nrm[:, 2] * ld[:, 2]).astype(f32)).astype(f32)
        diff = (f32(0.4) * np.maximum(ndl, f32(0))).astype(f32)
        light3 = (diff + f32(0.65)).astype(f32)
        c = (light3[:, None] * col).astype(f32)
    return np.minimum(np.maximum(c, f32(0)), f32(1))


def render_vispy(vertices, normals, colors01, faces, ob2cam, K, window, size=176, return_float=False, numpy_rule="numpy2", sub_bits=4):
    """The reference's VispyRenderer.render_image on this model: rgb uint8 [size,size,3], depth uint16 [size,size]
    (rows as the reference returns them)."""
    W = H = size
    P, V, light, (pA, pB) = vispy_uniforms(ob2cam, K, window)
    PV = _mat_mul_cols(P, V)
    v32 = np.asarray(vertices, np.float32)
    pv = project(clip_positions(v32, PV), W, H, sub_bits)
    zbuf, owner, setups = rasterize(pv, faces, W, H)
    colf = np.zeros((H, W, 3), f32)
    n32, c32 = np.asarray(normals, np.float32), np.asarray(colors01, np.float32)
    for t, s in setups.items():
        ys, xs = np.nonzero(owner == t)
        if len(ys) == 0:
            continue
        idx = s.idx
        attrs = np.concatenate([v32[idx], n32[idx], c32[idx]], 1)
        a = interpolate(s, attrs, xs, ys)
        colf[ys, xs] = shade_vispy(a[:, 0:3], a[:, 3:6], a[:, 6:9], light)
    rgb = unorm8(colf)
    rgb[owner < 0] = 0
    d16 = depth_mm(zbuf, pA, pB, numpy_rule)
    if return_float:
        return rgb, d16, colf, zbuf, owner
    return rgb, d16


def depth_mm(zbuf, A, B, numpy_rule="numpy2"):
    """vispy_renderer.py:163-169: float32 depth buffer -> uint16 millimetres.  A, B are float64 scalars there.
    'numpy2': `depth * -2.0 + 1.0` stays float32, `- A` promotes to float64, everything after is float64 (NEP 50);
    'numpy1': value-based casting -- the scalars are cast to float32, every operation is a float32 operation."""
    z = np.asarray(zbuf, f32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if numpy_rule == "numpy2":
            t = ((z * f32(-2.0)).astype(f32) + f32(1.0)).astype(f32).astype(np.float64)
            distance = np.float64(B) / (t - np.float64(A)) * -1.0
            distance[distance >= np.float64(B) / (np.float64(A) + 1)] = 0
            return (distance * 1000).astype(np.uint16)
        assert numpy_rule == "numpy1", numpy_rule
        t = (((z * f32(-2.0)).astype(f32) + f32(1.0)).astype(f32) - f32(A)).astype(f32)
        distance = ((f32(B) / t).astype(f32) * f32(-1)).astype(f32)
        distance[distance >= f32(np.float64(B) / (np.float64(A) + 1))] = 0
        return (distance * f32(1000)).astype(f32).astype(np.uint16)


def render_frame(vertices, colors01, faces, ob2cam, K, W, H, uv=None, texture=None, kd=(1.0, 1.0, 1.0), near=NEAR, far=FAR):
    """The second renderer's GL work (oracle/swiftshader_gl.py: render_frame_gl, this repo's statement of pyrender's scene) on this
    model: coverage, depth and vertex colours by the rules above (exact); the texture FILTER is plain float32 arithmetic
    (raster_oracle's bilinear / trilinear with the level of detail from the 2 x 2 quad's differences) -- the implementation
    filters in 16-bit fixed point, so textured colours agree to a few / 255 only.  rgb uint8 [H,W,3], depth uint16 [H,W]."""
    from . import raster_oracle as R
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * fx / W; P[1, 1] = 2.0 * fy / H
    P[0, 2] = 1.0 - 2.0 * cx / W; P[1, 2] = 2.0 * cy / H - 1.0
    P[2, 2] = (far + near) / (near - far); P[2, 3] = 2 * far * near / (near - far); P[3, 2] = -1.0
    V = np.diag([1.0, -1.0, -1.0, 1.0]).dot(np.asarray(ob2cam, np.float64))
    PV = (P @ V).astype(np.float32)
    v32 = np.asarray(vertices, np.float32)
    pv = project(clip_positions(v32, PV), W, H)
    zbuf, owner, setups = rasterize(pv, faces, W, H)
    colf = np.zeros((H, W, 3), f32)
    kd = np.asarray(kd, f32)
    levels = R.mip_pyramid(texture) if texture is not None else None
    for t, s in setups.items():
        ys, xs = np.nonzero(owner == t)
        if len(ys) == 0:
            continue
        if levels is None:
            base = interpolate(s, np.asarray(colors01, f32)[s.idx], xs, ys)
        else:
            uvt = np.asarray(uv, f32)[s.idx]
            th, tw = levels[0].shape[:2]
            c = interpolate(s, uvt, xs, ys)
            xq, yq = xs // 2 * 2, ys // 2 * 2
            q0, q1, q2 = interpolate(s, uvt, xq, yq), interpolate(s, uvt, xq + 1, yq), interpolate(s, uvt, xq, yq + 1)
            base = np.zeros((len(xs), 3), f32)
            for k in range(len(xs)):
                dx = ((q1[k] - q0[k]).astype(f32) * np.array([tw, th], f32)).astype(f32)
                dy = ((q2[k] - q0[k]).astype(f32) * np.array([tw, th], f32)).astype(f32)
                rho = max(float(np.sqrt(f32((dx * dx).astype(f32).sum(dtype=f32)))), float(np.sqrt(f32((dy * dy).astype(f32).sum(dtype=f32)))))
                lod = min(max(np.log2(max(rho, 1e-8)), 0.0), len(levels) - 1)
                l0 = int(np.floor(lod)); l1 = min(l0 + 1, len(levels) - 1); fl = f32(lod - l0)
                c0 = R._bilinear(levels[l0], c[k, 0], c[k, 1]); c1 = R._bilinear(levels[l1], c[k, 0], c[k, 1])
                base[k] = (c0 + fl * (c1 - c0)) / f32(255.0)
        colf[ys, xs] = np.minimum(np.maximum((base * kd[None, :]).astype(f32), f32(0)), f32(1))
    rgb = unorm8(colf)
    rgb[owner < 0] = 0
    rgb, z, hit = rgb[::-1].copy(), zbuf[::-1].copy(), (owner >= 0)[::-1].copy()       # rows flipped on read-back
    with np.errstate(divide="ignore", invalid="ignore"):
        z_ndc = (z * f32(2.0)).astype(f32) - f32(1.0)
        depth = (f32(2.0 * near * far) / (f32(far + near) - (z_ndc * f32(far - near)).astype(f32)).astype(f32)).astype(f32)
    depth[~hit] = 0
    return rgb, (depth * f32(1000)).astype(f32).astype(np.uint16)
