"""Seeded synthetic inputs shared by oracle/make_golden.py, tests/ and bench.py
(TEST INFRASTRUCTURE ONLY).  numpy's PCG64 and torch's CPU mt19937 streams are
platform-independent, so the GPU box regenerates exactly what the goldens were made from;
every golden file also stores fingerprints of its inputs to detect generator drift."""
import hashlib

import numpy as np
import torch

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
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((n, 4, res, res), generator=g) * scale
    B = torch.randn((n, 4, res, res), generator=g) * scale
    return A, B
