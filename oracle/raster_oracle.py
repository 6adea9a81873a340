"""CPU restatement (TEST INFRASTRUCTURE ONLY) of what the reference's OpenGL renderer computes for the
rendered image A: vispy_renderer.py:47-178 as driven by Tracker.render_window (predict.py:193-215).
Written as the literal GL pipeline -- the reference's own matrices (update_cam_mat :135-150), clip ->
NDC -> window transform, pixel-centre sampling with the top-left rule, LESS depth test, no face culling
(set_cull_face only selects glCullFace; GL_CULL_FACE is never enabled), perspective-correct varyings, the
fragment shader (:54-76), bottom-up read-back and the depth linearisation (:160-169).
Pinned (round 3) against a real OpenGL implementation: tests/golden/gl_swiftshader.npz holds what the UNMODIFIED reference class
VispyRenderer renders on SwiftShader's OpenGL ES 3.0 (oracle/swiftshader_gl.py, oracle/make_gl_golden.py); this file agrees with
it on coverage up to a handful of silhouette pixels (SwiftShader snaps vertices to 1/16 pixel), depth <= 1 mm, interior colours
<= 2-3 / 255 (tests/test_gl_swiftshader.py).  The geometry is additionally pinned against the analytic ray-sphere intersection
(tests/test_renderer.py: test_oracle_vs_analytic_sphere, and the same check on the HIP kernels)."""
import numpy as np


def projection_matrix(K, left, right, bottom, top, near=0.1, far=2.0):
    """update_cam_mat, vispy_renderer.py:135-150 (returns the un-transposed ortho . proj)."""
    proj = np.array([[K[0, 0], 0, -K[0, 2], 0], [0, K[1, 1], -K[1, 2], 0], [0, 0, near + far, near * far], [0, 0, -1, 0]])
    ortho = np.array([[2. / (right - left), 0, 0, -(right + left) / (right - left)],
                      [0, 2. / (top - bottom), 0, -(top + bottom) / (top - bottom)],
                      [0, 0, -2 / (far - near), -(far + near) / (far - near)], [0, 0, 0, 1]]).astype(np.float32)
    return ortho.dot(proj)


def render(vertices, normals, colors01, faces, ob2cam, K, window, res=176):
    """window = (left, top, right, bottom) of predict.py:203-206.  Returns rgb uint8 [res,res,3],
    depth uint16 [res,res]."""
    left, top, right, bottom = window
    P = projection_matrix(K, left, right, bottom, top).astype(np.float32)
    glcam_in_cvcam = np.diag([1.0, -1.0, -1.0, 1.0])
    V = np.linalg.inv(glcam_in_cvcam).dot(ob2cam)                       # predict.py:202
    light = np.dot(np.linalg.inv(V.T), np.array([0, 0.1, -0.9, 1]))[:3].astype(np.float32)  # :172
    PV = (P.astype(np.float64) @ V).astype(np.float32)
    vh = np.concatenate([vertices.astype(np.float32), np.ones((len(vertices), 1), np.float32)], 1)
    clip = vh @ PV.T
    w = clip[:, 3]
    ndc = clip[:, :3] / w[:, None]
    win = np.stack([(ndc[:, 0] + 1) * res / 2, (ndc[:, 1] + 1) * res / 2, (ndc[:, 2] + 1) / 2], 1).astype(np.float32)
    zbuf = np.ones((res, res), np.float32)
    rgb = np.zeros((res, res, 3), np.uint8)
    hit = np.zeros((res, res), bool)

    def edge(a, b, px, py):
        return (b[0] - a[0]) * (py - a[1]) - (b[1] - a[1]) * (px - a[0])

    def top_left(a, b):
        dx, dy = b[0] - a[0], b[1] - a[1]
        return dy < 0 or (dy == 0 and dx < 0)

    for t in range(len(faces)):
        ids = list(faces[t])
        if (w[ids] <= 0).any():
            continue
        v0, v1, v2 = win[ids[0]], win[ids[1]], win[ids[2]]
        area = np.float32(edge(v0, v1, v2[0], v2[1]))
        if area == 0 or np.isnan(area):
            continue
        if area < 0:
            ids[1], ids[2] = ids[2], ids[1]
            v1, v2 = v2, v1
            area = -area
        i0 = max(0, int(np.floor(min(v0[0], v1[0], v2[0]) - 0.5))); i1 = min(res - 1, int(np.ceil(max(v0[0], v1[0], v2[0]) - 0.5)))
        j0 = max(0, int(np.floor(min(v0[1], v1[1], v2[1]) - 0.5))); j1 = min(res - 1, int(np.ceil(max(v0[1], v1[1], v2[1]) - 0.5)))
        for j in range(j0, j1 + 1):
            for i in range(i0, i1 + 1):
                px, py = np.float32(i + 0.5), np.float32(j + 0.5)
                e0 = np.float32(edge(v1, v2, px, py)); e1 = np.float32(edge(v2, v0, px, py)); e2 = np.float32(edge(v0, v1, px, py))
                if not ((e0 > 0 or (e0 == 0 and top_left(v1, v2))) and (e1 > 0 or (e1 == 0 and top_left(v2, v0)))
                        and (e2 > 0 or (e2 == 0 and top_left(v0, v1)))):
                    continue
                ia = np.float32(1.0) / area
                l = np.array([e0 * ia, e1 * ia, e2 * ia], np.float32)
                zw = np.float32(l[0] * v0[2] + l[1] * v1[2] + l[2] * v2[2])
                if not (0 <= zw < 1) or not (zw < zbuf[j, i]):
                    continue
                zbuf[j, i] = zw
                hit[j, i] = True
                q = l * (np.float32(1.0) / w[ids]).astype(np.float32)
                b = (q / q.sum()).astype(np.float32)
                pos = b @ vertices[ids].astype(np.float32)
                nrm = b @ normals[ids].astype(np.float32)
                col = b @ colors01[ids].astype(np.float32)
                ld = -light - pos
                ld = ld / np.linalg.norm(ld)
                light3 = np.float32(0.4) * max(float(nrm @ ld), 0.0) + np.float32(0.65)
                rgb[j, i] = np.rint(np.clip(light3 * col, 0, 1) * 255).astype(np.uint8)
    # read-back: rows bottom-up, no flip (vispy_renderer.py:160-163); depth linearisation :164-169
    PT = P.T
    A, B = PT[2, 2], PT[3, 2]
    with np.errstate(divide="ignore"):
        distance = (B / (zbuf * np.float32(-2.0) + np.float32(1.0) - A) * -1).astype(np.float32)
    distance[distance >= B / (A + 1)] = 0
    distance[~hit] = 0
    return rgb, (distance * 1000).astype(np.uint16)


from .fixtures import icosphere  # noqa: E402,F401  (the test mesh lives with the other seeded fixtures)


# ------------------------------------------------------------------------------------------------------------------
# The reference's second renderer: offscreen_renderer.py:48-83 (pyrender), used for textured .obj models
# (predict.py:161-164, :209-213).  Restated as the GL pipeline pyrender sets up: IntrinsicsCamera projection at the
# camera's W x H, object pose = cvcam_in_glcam . ob_in_cvcam, a scene lit by ambient light [1,1,1] only (fragment
# colour = base colour = Kd x texture | vertex colour), colour rows flipped to top-down on read-back, depth buffer
# linearised to metres.  GL's sampling / fill rules pinned against SwiftShader (tests/test_gl_swiftshader.py); pyrender's own scene set-up
# (not installable offline) remains this file's reading.
# ------------------------------------------------------------------------------------------------------------------
def mip_pyramid(tex):
    """RGB uint8 [h,w,3] -> list of levels, each the 2x2 box filter of the previous (rounded to nearest)."""
    levels = [np.asarray(tex, np.uint8)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        t = levels[-1].astype(np.int32)
        h, w = t.shape[:2]
        nh, nw = max(h // 2, 1), max(w // 2, 1)
        ys0 = np.minimum(2 * np.arange(nh), h - 1); ys1 = np.minimum(2 * np.arange(nh) + 1, h - 1)
        xs0 = np.minimum(2 * np.arange(nw), w - 1); xs1 = np.minimum(2 * np.arange(nw) + 1, w - 1)
        s = t[ys0][:, xs0] + t[ys0][:, xs1] + t[ys1][:, xs0] + t[ys1][:, xs1]
        levels.append(((s + 2) >> 2).astype(np.uint8))
        if len(levels) >= 16:
            break
    return levels


def _bilinear(level, u, v):
    h, w = level.shape[:2]
    x = np.float32(u) * w - np.float32(0.5); y = (np.float32(1.0) - np.float32(v)) * h - np.float32(0.5)
    xf, yf = np.floor(x), np.floor(y)
    ax, ay = np.float32(x - xf), np.float32(y - yf)
    x0, y0 = int(xf) % w, int(yf) % h
    x1, y1 = (x0 + 1) % w, (y0 + 1) % h
    t = level.astype(np.float32)
    return (t[y0, x0] * (1 - ax) + t[y0, x1] * ax) * (1 - ay) + (t[y1, x0] * (1 - ax) + t[y1, x1] * ax) * ay


def render_frame(vertices, colors01, faces, ob2cam, K, W, H, uv=None, texture=None, kd=(1.0, 1.0, 1.0), near=0.1, far=2.0):
    """Returns rgb uint8 [H,W,3] and depth uint16 [H,W] = (pyrender depth * 1000).astype(uint16) (predict.py:211)."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * fx / W; P[1, 1] = 2.0 * fy / H
    P[0, 2] = 1.0 - 2.0 * cx / W; P[1, 2] = 2.0 * cy / H - 1.0
    P[2, 2] = (far + near) / (near - far); P[2, 3] = 2 * far * near / (near - far); P[3, 2] = -1.0
    V = np.diag([1.0, -1.0, -1.0, 1.0]).dot(ob2cam)         # cvcam_in_glcam . ob_in_cvcam (offscreen_renderer.py:80)
    PV = (P @ V).astype(np.float32)
    vh = np.concatenate([vertices.astype(np.float32), np.ones((len(vertices), 1), np.float32)], 1)
    clip = vh @ PV.T
    w = clip[:, 3]
    ndc = clip[:, :3] / w[:, None]
    win = np.stack([(ndc[:, 0] + 1) * W / 2, (ndc[:, 1] + 1) * H / 2, (ndc[:, 2] + 1) / 2], 1).astype(np.float32)
    zbuf = np.ones((H, W), np.float32)
    rgb = np.zeros((H, W, 3), np.uint8)
    hit = np.zeros((H, W), bool)
    levels = mip_pyramid(texture) if texture is not None else None
    kd = np.asarray(kd, np.float32)

    def edge(a, b, px, py):
        return (b[0] - a[0]) * (py - a[1]) - (b[1] - a[1]) * (px - a[0])

    def top_left(a, b):
        dx, dy = b[0] - a[0], b[1] - a[1]
        return dy < 0 or (dy == 0 and dx < 0)

    def bary(v0, v1, v2, area, px, py):
        e = np.array([edge(v1, v2, px, py), edge(v2, v0, px, py), edge(v0, v1, px, py)], np.float32)
        return e, e * (np.float32(1.0) / area)

    for t in range(len(faces)):
        ids = list(faces[t])
        if (w[ids] <= 0).any():
            continue
        v0, v1, v2 = win[ids[0]], win[ids[1]], win[ids[2]]
        area = np.float32(edge(v0, v1, v2[0], v2[1]))
        if area == 0 or np.isnan(area):
            continue
        if area < 0:
            ids[1], ids[2] = ids[2], ids[1]
            v1, v2 = v2, v1
            area = -area
        i0 = max(0, int(np.floor(min(v0[0], v1[0], v2[0]) - 0.5))); i1 = min(W - 1, int(np.ceil(max(v0[0], v1[0], v2[0]) - 0.5)))
        j0 = max(0, int(np.floor(min(v0[1], v1[1], v2[1]) - 0.5))); j1 = min(H - 1, int(np.ceil(max(v0[1], v1[1], v2[1]) - 0.5)))
        iw = (np.float32(1.0) / w[ids]).astype(np.float32)
        for j in range(j0, j1 + 1):
            for i in range(i0, i1 + 1):
                px, py = np.float32(i + 0.5), np.float32(j + 0.5)
                e, l = bary(v0, v1, v2, area, px, py)
                if not ((e[0] > 0 or (e[0] == 0 and top_left(v1, v2))) and (e[1] > 0 or (e[1] == 0 and top_left(v2, v0)))
                        and (e[2] > 0 or (e[2] == 0 and top_left(v0, v1)))):
                    continue
                zw = np.float32(l[0] * v0[2] + l[1] * v1[2] + l[2] * v2[2])
                if not (0 <= zw < 1) or not (zw < zbuf[j, i]):
                    continue
                zbuf[j, i] = zw
                hit[j, i] = True
                if levels is not None:
                    uvs = []
                    for (ox, oy) in ((0, 0), (1, 0), (0, 1)):
                        _, m = bary(v0, v1, v2, area, px + np.float32(ox), py + np.float32(oy))
                        q = m * iw
                        uvs.append((q / q.sum()).astype(np.float32) @ uv[ids].astype(np.float32))
                    th, tw = levels[0].shape[:2]
                    dx = (uvs[1] - uvs[0]) * np.array([tw, th], np.float32)
                    dy = (uvs[2] - uvs[0]) * np.array([tw, th], np.float32)
                    rho = max(float(np.sqrt((dx * dx).sum())), float(np.sqrt((dy * dy).sum())))
                    lod = min(max(np.log2(max(rho, 1e-8)), 0.0), len(levels) - 1)
                    l0 = int(np.floor(lod)); l1 = min(l0 + 1, len(levels) - 1); fl = np.float32(lod - l0)
                    c0 = _bilinear(levels[l0], uvs[0][0], uvs[0][1]); c1 = _bilinear(levels[l1], uvs[0][0], uvs[0][1])
                    col = (c0 + fl * (c1 - c0)) / np.float32(255.0)
                else:
                    q = l * iw
                    col = (q / q.sum()).astype(np.float32) @ colors01[ids].astype(np.float32)
                rgb[j, i] = np.rint(np.clip(col * kd, 0, 1) * 255).astype(np.uint8)
    # window rows count bottom-up from Y = H - v: row j of this buffer already is image row j of the mapping used above?
    # NO: here win y = (y_ndc + 1) H / 2 with y_ndc = 1 - 2 v / H  ->  win y = H - v: bottom-up.  pyrender flips on read-back.
    rgb = rgb[::-1].copy(); zbuf = zbuf[::-1].copy(); hit = hit[::-1].copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        z_ndc = zbuf * np.float32(2.0) - np.float32(1.0)
        depth = (np.float32(2.0 * near * far) / (np.float32(far + near) - z_ndc * np.float32(far - near))).astype(np.float32)
    depth[~hit] = 0
    return rgb, (depth * 1000).astype(np.uint16)
