"""Deterministic synthetic dataset / result trees in the reference's on-disk layouts (TEST INFRASTRUCTURE ONLY), used to
cross-check the product's evaluators (iros20-6d-pose-tracking_amd/sequence.py, metrics.py) against the reference's own
eval_ycb.eval_one_class (eval_ycb.py:67-119) and eval_ycbineoat.eval_all (eval_ycbineoat.py:49-109).

The RESULT trees are written by the product's own drivers (sequence.get_results_ycb / predict_sequence_ycbineoat) with a
stub tracker, so that the file layout the reference evaluator parses (seq<ID>/%07d.txt, file index = frame id - 1; one folder
per YCBInEOAT video with %07d.txt) is the layout the product really writes -- not this file's reading of it."""
import os

import numpy as np
from PIL import Image
from scipy.spatial.transform import Rotation

YCB_CLASSES = ("002_master_chef_can", "003_cracker_box", "004_sugar_box", "005_tomato_soup_can", "006_mustard_bottle",
               "021_bleach_cleanser")
# (sequence id, frames); 0010 is a training video (skipped by the driver, predict.py:349), 0055 lacks the class
YCB_SEQUENCES = ((48, 40), (50, 25), (59, 30), (10, 6))
EOAT_VIDEOS = (("cracker_box_reorient", 18), ("mustard0", 22), ("bleach_hard_00_03_chaitanya", 15), ("sugar_box1", 12),
               ("tomato_soup_can_yalehand0", 9))


class StubTracker:
    """on_track(prev_pose, rgb, depth, ...) -> prev_pose composed with a seeded small motion + drift: stands where
    se3tracknet_amd.Tracker stands in the sequence drivers (no GPU).  Errors grow along a sequence from sub-millimetre to
    beyond the 0.1 m AUC cap, so every branch of VOCap (ties at 0, values above the cap) is exercised."""

    def __init__(self, seed, step_m=0.004, step_deg=1.5):
        self.rng = np.random.default_rng(seed)
        self.step_m, self.step_deg = step_m, step_deg
        self.calls = 0

    def on_track(self, prev_pose, rgb, depth, **kw):
        assert rgb.ndim == 3 and rgb.dtype == np.uint8 and depth.dtype == np.uint16
        self.calls += 1
        D = np.eye(4)
        D[:3, :3] = Rotation.from_rotvec(self.rng.normal(0, np.deg2rad(self.step_deg), 3)).as_matrix()
        D[:3, 3] = self.rng.normal(0, self.step_m, 3) + np.array([0.0015, 0.0, 0.001])
        return D @ np.asarray(prev_pose, np.float64)


def _model_points(seed, n=400):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, (n, 3)) * np.array([0.04, 0.06, 0.1])


def _gt_pose(seq, i):
    P = np.eye(4)
    P[:3, :3] = Rotation.from_rotvec([0.3 + 0.01 * i, -0.2 + 0.004 * seq, 0.1 * np.sin(0.1 * i)]).as_matrix()
    P[:3, 3] = (0.05 * np.sin(0.05 * i), -0.02 + 0.001 * i, 0.8 + 0.002 * i)
    return P


def _tiny_frame(path_rgb, path_depth, seed):
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)).save(path_rgb)
    Image.fromarray(rng.integers(400, 1200, (12, 16)).astype(np.uint16)).save(path_depth)


def make_ycb_tree(root, class_id=2):
    """<root>/ycb/{CADmodels/*/points.xyz, YCB_Video_toolbox/keyframe.txt, data_organized/%04d/{color,depth_filled,pose_gt/<id>}}.
    Returns the ycb_dir."""
    ycb = os.path.join(root, "ycb")
    for k, name in enumerate(YCB_CLASSES):
        os.makedirs(os.path.join(ycb, "CADmodels", name), exist_ok=True)
        np.savetxt(os.path.join(ycb, "CADmodels", name, "points.xyz"), _model_points(10 + k))
    os.makedirs(os.path.join(ycb, "YCB_Video_toolbox"), exist_ok=True)
    keys = []
    for seq, n in YCB_SEQUENCES + ((55, 5),):
        sdir = os.path.join(ycb, "data_organized", "%04d" % seq)
        for d in ("color", "depth_filled"):
            os.makedirs(os.path.join(sdir, d), exist_ok=True)
        if seq != 55:
            os.makedirs(os.path.join(sdir, "pose_gt", str(class_id)), exist_ok=True)
        else:
            os.makedirs(os.path.join(sdir, "pose_gt", str(class_id + 1)), exist_ok=True)
        for i in range(n):
            fid = i + 1                                   # YCB-Video frame ids start at 1
            _tiny_frame(os.path.join(sdir, "color", "%06d.png" % fid), os.path.join(sdir, "depth_filled", "%06d.png" % fid),
                        1000 * seq + i)
            if seq != 55:
                np.savetxt(os.path.join(sdir, "pose_gt", str(class_id), "%06d.txt" % fid), _gt_pose(seq, i))
            if fid % 3 == 1 or fid == n:                  # keyframes: every third frame and the last one
                keys.append("%04d/%06d" % (seq, fid))
    keys += ["0048/000999", "0060/000001"]               # keyframes without a result file are simply never visited
    with open(os.path.join(ycb, "YCB_Video_toolbox", "keyframe.txt"), "w") as f:
        f.write("\n".join(keys) + "\n")
    return ycb


def make_ycb_results(sequence_module, ycb_dir, out_dir, class_id=2):
    """The product's own getResultsYcb driver writes the result tree (stub tracker)."""
    return sequence_module.get_results_ycb(StubTracker(5), ycb_dir, class_id, out_dir)


def make_eoat_tree(root):
    """<root>/eoat/<video>/{rgb,depth_filled,annotated_poses}.  Returns the data_dir."""
    data = os.path.join(root, "eoat")
    for v, (name, n) in enumerate(EOAT_VIDEOS):
        for d in ("rgb", "depth_filled", "annotated_poses"):
            os.makedirs(os.path.join(data, name, d), exist_ok=True)
        for i in range(n):
            _tiny_frame(os.path.join(data, name, "rgb", "%07d.png" % i), os.path.join(data, name, "depth_filled", "%07d.png" % i),
                        77000 + 100 * v + i)
            np.savetxt(os.path.join(data, name, "annotated_poses", "%07d.txt" % i), _gt_pose(70 + v, i))
    return data


def make_eoat_results(sequence_module, data_dir, res_dir):
    out = {}
    for v, (name, n) in enumerate(EOAT_VIDEOS):
        # a slower drift for the later videos: their AUCs differ
        trk = StubTracker(40 + v, step_m=0.004 / (1 + v), step_deg=1.5 / (1 + v))
        out[name] = sequence_module.predict_sequence_ycbineoat(trk, os.path.join(data_dir, name), os.path.join(res_dir, name))
    return out
