"""CPU restatement (TEST INFRASTRUCTURE ONLY) of what the reference's OpenGL renderer computes for the
rendered image A: vispy_renderer.py:47-178 as driven by Tracker.render_window (predict.py:193-215).
Written as the literal GL pipeline -- the reference's own matrices (update_cam_mat :135-150), clip ->
NDC -> window transform, pixel-centre sampling with the top-left rule, LESS depth test, no face culling
(set_cull_face only selects glCullFace; GL_CULL_FACE is never enabled), perspective-correct varyings, the
fragment shader (:54-76), bottom-up read-back and the depth linearisation (:160-169).
PARITY UNPINNED: no OpenGL implementation is available offline to compare against."""
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


def icosphere(subdiv=2, radius=0.05, seed=0):
    """Test mesh: subdivided icosahedron, outward (CCW) faces, random vertex colours, analytic normals."""
    t = (1 + 5 ** 0.5) / 2
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, float) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (v[a] + v[b]) / 2
                v.append(m / np.linalg.norm(m)); cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    v = np.array(v)
    rng = np.random.default_rng(seed)
    return dict(vertices=(v * radius).astype(np.float32), faces=np.array(f, np.int32),
                colors=rng.integers(40, 256, (len(v), 3)).astype(np.float64), normals=v.copy())
