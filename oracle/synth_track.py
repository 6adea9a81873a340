"""A synthetic tracking problem WITH ground truth (stand-in for BASELINE configs[2], YCB-Video seq 0048 + pretrained weights:
neither is available offline).  TEST INFRASTRUCTURE ONLY: fixtures + the data the stand-in weights were trained on
(scripts/train_synth_tracker.py); nothing here is imported by the product package.

  object      an ellipsoid (semi-axes 60 / 45 / 35 mm) with a smooth vertex-colour pattern: rotation and translation are both
              observable in RGB-D, unlike the random-colour sphere of oracle/closed_loop.py
  trajectory  ground-truth pose G_f: translation on closed_loop.anchor (4-7 mm per frame), rotation a seeded smooth curve
              (3-4.5 degrees per frame) -- inside the 0.03 m / 30 degree normalisers of predict.py:586 (the YCBInEOAT regime)
  frame f     a structured 480x640 RGB-D background (fixtures.structured_frame, 16 distinct, cycled) with the object rendered
              at G_f pasted in: the reference's renderer (oracle/ss_fast.py) at the native resolution of the crop window
  sample      (image A rendered at a perturbed pose P_A, the frame cropped at P_A's window as predict.py:236-262 does, labels
              trans = (t_G - t_A) / trans_normalizer, rot = rotvec(R_G R_A^T) / rot_normalizer -- datasets.py:138-150)

With weights trained on such samples the loop of predict.py:416-420 is CONTRACTIVE: the network re-estimates the pose from the
observed frame every step, so a rounding difference in one frame's estimate is corrected by the next frame instead of being
amplified (the random-init loop of oracle/closed_loop.py amplifies it: profiles/r06_free_run.json)."""
import os

import numpy as np

from . import closed_loop as CL
from . import fixtures as Fx
from . import se3_oracle as O

RADII = np.array([0.060, 0.045, 0.035])
OBJECT_WIDTH_MM = CL.OBJECT_WIDTH_MM
REGIME = os.environ.get("SE3TN_SYNTH_REGIME", "ycbineoat_30deg")   # the default regime of this module; predict.py:586: 0.03 m, 30 degrees: the regime where a logit error reaches the pose x 0.52 and the
                                    # 1e-5 pose tolerance binds; rotations of up to 27 degrees between image A and the frame are also what
                                    # lets a rotation head learn from 320 k synthetic pairs (under 5 degrees it did not start to)
TRANS_NORMALIZER, ROT_NORMALIZER = CL.REGIMES[REGIME]


def normalizers(regime=None):
    return CL.REGIMES[regime or REGIME]


def rot_speed(regime=None):
    """ground-truth rotation per frame scales with the normaliser: 3-4.5 degrees per frame under 30 degrees, 1-1.5 under 5"""
    return 1.0 if (regime or REGIME) == "ycbineoat_30deg" else 1.0 / 3.0
N_BACKGROUNDS = CL.N_DISTINCT_FRAMES


def make_object(subdiv=4):
    """ellipsoid mesh: vertices float32 [V,3] (metres), analytic normals, colours 0..255 (float64, as fixtures.icosphere)"""
    m = Fx.icosphere(subdiv, 1.0, 0)
    d = np.asarray(m["normals"], np.float64)                       # unit directions
    v = d * RADII
    n = d / RADII
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    col = np.stack([0.5 + 0.33 * x + 0.15 * np.sin(7 * y + 1.0), 0.5 + 0.33 * y + 0.15 * np.sin(6 * z + 2.0),
                    0.5 + 0.33 * z + 0.15 * np.sin(8 * x + 0.5)], 1)
    col = np.round(np.clip(col, 0.05, 1.0) * 255)
    return dict(vertices=v.astype(np.float32), faces=np.asarray(m["faces"], np.int32), colors=col.astype(np.float64), normals=n)


def rodrigues64(r):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(np.asarray(r, np.float64)).as_matrix()


def gt_pose(seed, f, regime=None):
    """ground-truth object-in-camera pose of frame f of sequence `seed`"""
    rng = np.random.default_rng(7000 + seed)
    R0 = rodrigues64(rng.normal(0, 0.8, 3))
    ph = rng.uniform(0, 2 * np.pi, 3)
    w = rot_speed(regime) * np.array([1.20 * np.sin(2 * np.pi * f / 173.0 + ph[0]), 0.90 * np.sin(2 * np.pi * f / 211.0 + ph[1]),
                              1.35 * np.sin(2 * np.pi * f / 139.0 + ph[2])])
    P = np.eye(4)
    P[:3, :3] = rodrigues64(w) @ R0
    P[:3, 3] = CL.anchor(f + 37 * seed)
    return P


def crop_window_of(P, K):
    bb = O.compute_bbox(P, K, OBJECT_WIDTH_MM, scale=(1000, 1000, 1000))
    return int(bb[:, 1].min()), int(bb[:, 0].min()), int(bb[:, 1].max()), int(bb[:, 0].max())     # left, top, right, bottom


def object_patch(om, G, K, numpy_rule="numpy1"):
    """the object at pose G rendered over its own crop window at that window's native resolution (one render pixel per frame
    pixel): returns (left, top, rgb [s,s,3] uint8, depth [s,s] uint16 mm, 0 = background)"""
    from . import ss_fast as SF
    l, t, r, b = crop_window_of(G, K)
    bbf = O.compute_bbox(G, K, OBJECT_WIDTH_MM, scale=(1000, -1000, 1000))
    win = (int(bbf[:, 1].min()), int(bbf[:, 0].min()), int(bbf[:, 1].max()), int(bbf[:, 0].max()))
    s = max(r - l, b - t, 8)
    rgb, dep = SF.render_vispy(om[0], om[1], om[2], om[3], G, K, win, size=s, numpy_rule=numpy_rule)
    return l, t, rgb[:b - t, :r - l], dep[:b - t, :r - l]


def compose_frame(background, patch):
    """paste the object into a copy of the background frame (no occlusion reasoning: the object is in front)"""
    rgb, depth = background[0].copy(), background[1].copy()
    l, t, prgb, pdep = patch
    H, W = depth.shape
    y0, x0 = max(t, 0), max(l, 0)
    y1, x1 = min(t + pdep.shape[0], H), min(l + pdep.shape[1], W)
    if y1 > y0 and x1 > x0:
        sub = pdep[y0 - t:y1 - t, x0 - l:x1 - l]
        m = sub > 0
        rgb[y0:y1, x0:x1][m] = prgb[y0 - t:y1 - t, x0 - l:x1 - l][m]
        depth[y0:y1, x0:x1][m] = sub[m]
    return rgb, depth


def backgrounds():
    return [Fx.structured_frame(400 + i) for i in range(N_BACKGROUNDS)]


def sequence_patches(job):
    """worker: object patches of frames [f0, f1) of sequence `seed`"""
    om = CL.oracle_mesh(make_object(job.get("subdiv", 4)))
    K = np.asarray(job["K"], np.float64)
    return [object_patch(om, gt_pose(job["seed"], f, job.get("regime")), K) for f in range(job["f0"], job["f1"])]


class Sequence:
    """frames of one synthetic sequence, composed on demand from the cached backgrounds and the per-frame object patches"""

    def __init__(self, seed, patches, offset=None):
        self.seed, self.patches = seed, patches
        self.bg = backgrounds()
        self.offset = 5 * seed if offset is None else offset

    def __len__(self):
        return len(self.patches)

    def frame(self, f):
        return compose_frame(self.bg[(f + self.offset) % N_BACKGROUNDS], self.patches[f])

    def save(self, path):
        arrs = {}
        for f, (l, t, rgb, dep) in enumerate(self.patches):
            arrs["lt_%d" % f] = np.array([l, t], np.int32)
            arrs["rgb_%d" % f] = rgb
            arrs["dep_%d" % f] = dep
        np.savez(path, n=len(self.patches), seed=self.seed, offset=self.offset, **arrs)

    @staticmethod
    def load(path):
        z = np.load(path)
        patches = []
        for f in range(int(z["n"])):
            lt = z["lt_%d" % f]
            patches.append((int(lt[0]), int(lt[1]), z["rgb_%d" % f], z["dep_%d" % f]))
        return Sequence(int(z["seed"]), patches, int(z["offset"]))


def make_sequence(seed, frames, K, pool=None, chunk=50, subdiv=4, regime=None):
    jobs = [dict(seed=seed, f0=f0, f1=min(f0 + chunk, frames), K=K, subdiv=subdiv, regime=regime) for f0 in range(0, frames, chunk)]
    parts = list(pool.map(sequence_patches, jobs)) if pool is not None else [sequence_patches(j) for j in jobs]
    return Sequence(seed, [p for part in parts for p in part])


# ------------------------------------------------------------------------------------------------------------------
# training samples (scripts/train_synth_tracker.py)
# ------------------------------------------------------------------------------------------------------------------
def random_gt(rng):
    P = np.eye(4)
    P[:3, :3] = rodrigues64(rng.normal(0, 1.2, 3))
    P[:3, 3] = (rng.uniform(-0.085, 0.085), rng.uniform(-0.055, 0.055), rng.uniform(0.62, 0.98))
    return P


def perturbed(G, rng, scale=None, regime=None):
    """P_A and the labels such that processPredict(P_A, labels) = G (datasets.py:159-175): t_G = t_A + trans * tn,
    R_G = Rodrigues(rot * rn) R_A"""
    s = rng.uniform(0.05, 1.0) if scale is None else scale
    trans = rng.uniform(-0.9, 0.9, 3) * s
    rot = rng.normal(0, 1, 3)
    rot = rot / np.linalg.norm(rot) * rng.uniform(0, 0.9) * s
    A = np.eye(4)
    tn, rn = normalizers(regime)
    A[:3, 3] = G[:3, 3] - trans * tn
    A[:3, :3] = rodrigues64(rot * rn).T @ G[:3, :3]
    return A, trans.astype(np.float32), rot.astype(np.float32)


def training_samples(job):
    """worker: `n` samples as raw crops (uint8 / uint16, before OffsetDepth / normalisation) + poses + labels"""
    rng = np.random.default_rng(90000 + job["seed"])
    om = CL.oracle_mesh(make_object(job.get("subdiv", 4)))
    K = np.asarray(job["K"], np.float64)
    bgs = [Fx.structured_frame(1000 + (job["seed"] * 7 + i) % 97) for i in range(4)]
    n = job["n"]
    out = dict(rgbA=np.empty((n, 176, 176, 3), np.uint8), depthA=np.empty((n, 176, 176), np.uint16),
               rgbB=np.empty((n, 176, 176, 3), np.uint8), depthB=np.empty((n, 176, 176), np.uint16),
               zA=np.empty(n, np.float64), trans=np.empty((n, 3), np.float32), rot=np.empty((n, 3), np.float32))
    for i in range(n):
        G = random_gt(rng)
        A, trans, rot = perturbed(G, rng, regime=job.get("regime"))
        rgb, depth = compose_frame(bgs[i % len(bgs)], object_patch(om, G, K))
        rgbA, depthA = CL.oracle_image_A(om, A, K, OBJECT_WIDTH_MM, "numpy1")
        bb = O.compute_bbox(A, K, OBJECT_WIDTH_MM, scale=(1000, 1000, 1000))
        rgbB, depthB = O.crop_bbox(rgb, depth, bb, (176, 176))
        out["rgbA"][i], out["depthA"][i], out["rgbB"][i], out["depthB"][i] = rgbA, depthA, rgbB, depthB
        out["zA"][i], out["trans"][i], out["rot"][i] = A[2, 3], trans, rot
    return out


def offset_depth_torch(depth_u16, zA, torch):
    """OffsetDepth (data_augmentation.py:134-144) on a batch, NumPy-1 rule (float32 arithmetic), as torch tensors"""
    d = depth_u16.to(torch.float32)
    invalid = (d <= 100) | (d >= 2000)
    d = d - (zA.to(torch.float32) * 1000).reshape(-1, 1, 1)
    d[invalid] = 2000.0
    return d
