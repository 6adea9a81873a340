"""Seeded synthetic inputs shared by oracle/make_golden.py, tests/ and bench.py
(TEST INFRASTRUCTURE ONLY).  numpy's PCG64 and torch's CPU mt19937 streams are
platform-independent, so the GPU box regenerates exactly what the goldens were made from;
every golden file also stores fingerprints of its inputs to detect generator drift."""
import hashlib

import numpy as np

# dataset_info.yml:4-7
K_YCB = np.array([[1066.778, 0.0, 312.9869], [0.0, 1067.487, 241.3109], [0.0, 0.0, 1.0]])
DATASET_INFO = {
    "camera": {"height": 480, "width": 640, "focalX": 1066.778, "focalY": 1067.487,
               "centerX": 312.9869, "centerY": 241.3109},
    "resolution": 176, "boundingbox": 10, "object_width": 250.0,
}


def sha(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def synthetic_frame(seed, H=480, W=640):
    """rgb u8 [H,W,3], depth u16 [H,W] in mm with holes (0), near (<=100) and far (>=2000)."""
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    depth = rng.integers(500, 1200, (H, W)).astype(np.uint16)
    sel = rng.random((H, W))
    depth[sel < 0.05] = 0
    depth[(sel >= 0.05) & (sel < 0.07)] = rng.integers(1, 101, int(((sel >= 0.05) & (sel < 0.07)).sum()))
    depth[(sel >= 0.07) & (sel < 0.10)] = rng.integers(2000, 5000, int(((sel >= 0.07) & (sel < 0.10)).sum()))
    return rgb, depth


def structured_frame(seed, H=480, W=640):
    """Camera frame with image structure (low-frequency colour fields, per-frame gain / offset / noise level, a tilted depth
    plane with bumps and holes): unlike synthetic_frame's i.i.d. noise, the network's pooled features -- and so its
    (trans, rot) output -- differ from frame to frame and with the crop window (oracle/closed_loop.py)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    rgb = np.zeros((H, W, 3))
    for c in range(3):
        f = np.zeros((H, W))
        for _ in range(4):
            fx, fy = rng.uniform(-0.04, 0.04, 2)
            f += rng.uniform(0.3, 1.0) * np.sin(fx * xx + fy * yy + rng.uniform(0, 2 * np.pi))
        rgb[..., c] = f
    rgb = (rgb - rgb.min()) / (rgb.max() - rgb.min())
    gain = rng.uniform(80, 255)
    rgb = rgb * gain + rng.uniform(0, 255 - gain) + rng.normal(0, rng.uniform(2, 25), (H, W, 3))
    rgb = np.clip(rgb, 0, 255).astype(np.uint8)
    d = rng.uniform(600, 1000) + rng.uniform(-0.4, 0.4) * (xx - W / 2) + rng.uniform(-0.4, 0.4) * (yy - H / 2)
    for _ in range(5):
        cx, cy, r = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(30, 120)
        d -= rng.uniform(50, 250) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
    d += rng.normal(0, 3, (H, W))
    depth = np.clip(d, 300, 2500).astype(np.uint16)
    depth[rng.random((H, W)) < 0.04] = 0
    return rgb, depth


def synthetic_render(seed, z_m, res=176):
    """Stand-in for Tracker.render_window (predict.py:193-215): rgbA u8 [res,res,3],
    depthA u16 [res,res] mm, background exactly 0, object a disc around depth z."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:res, 0:res]
    obj = (yy - res / 2) ** 2 + (xx - res / 2) ** 2 < (0.38 * res) ** 2
    rgb = rng.integers(0, 256, (res, res, 3), dtype=np.uint8) * obj[..., None].astype(np.uint8)
    depth = (z_m * 1000 + rng.integers(-60, 60, (res, res))).astype(np.uint16) * obj.astype(np.uint16)
    return rgb, depth


def mean_std(seed):
    """mean.npy / std.npy surface: float64 [8] = A(R,G,B,D) then B(R,G,B,D)
    (predict.py:657-658, train.py:114-125)."""
    rng = np.random.default_rng(seed)
    mean = rng.uniform(50, 150, 8)
    std = rng.uniform(10, 60, 8)
    return mean, std


def pose(seed, t=(0.05, -0.02, 0.8)):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    P = np.eye(4)
    P[:3, :3] = Rotation.from_rotvec(rng.normal(0, 0.6, 3)).as_matrix()
    P[:3, 3] = t
    return P


def net_inputs(seed, n, scale=1.0, res=176):
    import torch                                   # lazily: the renderer fixtures also run under an interpreter without torch
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((n, 4, res, res), generator=g) * scale
    B = torch.randn((n, 4, res, res), generator=g) * scale
    return A, B


def depth_frame_with_holes(seed, H=120, W=160, hole_frac=0.25):
    """uint16 mm depth with holes: random blobs of zeros, a few near (<100 mm) and far (> 2 m) pixels, a large empty corner
    (fill_depth fixtures: tests/test_fill_depth.py, oracle/pin_opencv.py)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    d = (700 + 150 * np.sin(xx / 17.0) + 100 * np.cos(yy / 11.0) + rng.integers(-8, 9, (H, W))).astype(np.float64)
    holes = rng.random((H, W)) < hole_frac * 0.3
    for _ in range(12):
        cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(2, 9)
        holes |= (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    d[holes] = 0
    d[rng.random((H, W)) < 0.01] = rng.integers(1, 100)
    d[rng.random((H, W)) < 0.01] = rng.integers(2100, 4000)
    d[: H // 6, : W // 5] = 0                      # a large empty corner (reaches the image border)
    return d.astype(np.uint16)


def depth_frame_with_far_wall(seed):
    """+ a 40 x 60 region at 2.5-3 m (beyond max_depth = 2 m), larger than every structuring element of fill_depth."""
    mm = depth_frame_with_holes(seed, 240, 320)
    rng = np.random.default_rng(seed)
    mm[100:140, 200:260] = rng.integers(2500, 3000, (40, 60))
    return mm


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


def textured_sphere(subdiv=2, radius=0.05, tex_hw=(64, 128)):
    """icosphere with spherical texture coordinates and a synthetic RGB texture (ramps + checker + white speckles): the mesh of the
    pyrender-route tests (tests/test_renderer.py, tests/test_gl_swiftshader.py).  Returns dict(vertices f64, faces, uv, texture u8, kd)."""
    m = icosphere(subdiv, radius, 3)
    v = m["vertices"].astype(np.float64)
    n = v / np.linalg.norm(v, axis=1, keepdims=True)
    uv = np.stack([0.5 + np.arctan2(n[:, 1], n[:, 0]) / (2 * np.pi), 0.5 + np.arcsin(np.clip(n[:, 2], -1, 1)) / np.pi], 1)
    rng = np.random.default_rng(9)
    th, tw = tex_hw
    yy, xx = np.mgrid[0:th, 0:tw]
    tex = np.stack([(xx * 255 // (tw - 1)), (yy * 255 // (th - 1)), ((xx // 8 + yy // 8) % 2) * 200 + 30], -1).astype(np.uint8)
    tex[rng.random((th, tw)) < 0.05] = (255, 255, 255)
    return dict(vertices=v, faces=m["faces"], uv=uv, texture=tex, kd=np.array([0.9, 1.0, 0.8]), colors=m["colors"])
