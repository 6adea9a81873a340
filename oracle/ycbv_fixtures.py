"""A deterministic synthetic YCB-Video tree in the layout the reference's YCB-Video drivers read (TEST INFRASTRUCTURE ONLY):
predictSequenceYcb (predict.py:446-575), getResultsYcb (:299-443), use_posecnn_res (:88-123) and eval_ycb.eval_one_class.

    <ycb>/data_organized/%04d/{color,depth_filled}/%06d.png   frame ids start at 1; uint8 RGB / uint16 millimetres
    <ycb>/data_organized/%04d/pose_gt/<class>/%06d.txt        4x4, np.savetxt
    <ycb>/image_sets/keyframe.txt                             'SSSS/FFFFFF' per line (use_posecnn_res, getResultsYcb)
    <ycb>/YCB_Video_toolbox/keyframe.txt                      the same list (eval_ycb.eval_one_class)
    <ycb>/YCB_Video_toolbox/results_PoseCNN_RSS2018/%06d.mat  one per keyframe LINE INDEX: rois [k,6] (column 1 = class id),
                                                              poses_icp [k,7] = quaternion (w,x,y,z) + translation
    <ycb>/CADmodels/<name>/points.xyz                         model points per class, sorted by name

Sequences: 0048 and 0050 (test set, contain the class), 0049 (test set, another class only), 0010 (training video with the
class: getResultsYcb must skip it, predict.py:349).  Everything is seeded; tests regenerate the tree instead of storing it."""
import os

import numpy as np
from PIL import Image
from scipy.io import savemat
from scipy.spatial.transform import Rotation

from . import fixtures as Fx

CLASS_ID = 2
OTHER_CLASS = 5
FRAME_HW = (240, 320)
SEQUENCES = ((10, 3, True), (48, 9, True), (49, 4, False), (50, 6, True))      # (id, frames, contains CLASS_ID)
KEYFRAMES = {10: (1, 3), 48: (1, 4, 7, 9), 49: (2,), 50: (1, 3, 5)}               # frame ids (1-based)
CLASS_NAMES = ("002_master_chef_can", "003_cracker_box", "004_sugar_box", "005_tomato_soup_can", "006_mustard_bottle")
REINIT_FRAMES = "0048/000005,0048/000008"                                       # --reinit_frames of the ycbv-mode run


def gt_pose(seq, i):
    """object-in-camera pose of frame index i (0-based) of sequence `seq`: a smooth track in front of the 240 x 320 frames"""
    P = np.eye(4)
    P[:3, :3] = Rotation.from_rotvec([0.4 + 0.03 * i, -0.3 + 0.01 * (seq - 48), 0.2 * np.sin(0.4 * i)]).as_matrix()
    P[:3, 3] = (-0.08 + 0.003 * i, -0.062 + 0.002 * (seq - 48), 0.55 + 0.004 * i)
    return P


def keyframe_lines():
    return ["%04d/%06d" % (s, f) for s in sorted(KEYFRAMES) for f in KEYFRAMES[s]]


def posecnn_pose(seq, frame_id, class_id):
    """what the PoseCNN result file of keyframe (seq, frame_id) holds for `class_id`: the ground truth disturbed by a seeded
    centimetre / few-degree error -> (quaternion wxyz, translation)"""
    rng = np.random.default_rng(7919 * seq + 31 * frame_id + class_id)
    P = gt_pose(seq, frame_id - 1)
    R = Rotation.from_rotvec(rng.normal(0, np.deg2rad(3.0), 3)).as_matrix() @ P[:3, :3]
    t = P[:3, 3] + rng.normal(0, 0.006, 3)
    q = Rotation.from_matrix(R).as_quat()                      # scipy: (x, y, z, w)
    return np.r_[q[3], q[:3]], t


def make_tree(root):
    """writes the tree under <root>/ycb and returns that path"""
    ycb = os.path.join(root, "ycb")
    rng = np.random.default_rng(5)
    for k, name in enumerate(CLASS_NAMES):
        os.makedirs(os.path.join(ycb, "CADmodels", name), exist_ok=True)
        np.savetxt(os.path.join(ycb, "CADmodels", name, "points.xyz"), Fx.icosphere(2, 0.05 + 0.005 * k, 40 + k)["vertices"])
    for seq, n, has in SEQUENCES:
        sdir = os.path.join(ycb, "data_organized", "%04d" % seq)
        classes = (CLASS_ID, OTHER_CLASS) if has else (OTHER_CLASS,)
        for d in ("color", "depth_filled") + tuple(os.path.join("pose_gt", str(c)) for c in classes):
            os.makedirs(os.path.join(sdir, d), exist_ok=True)
        for i in range(n):
            rgb, depth = Fx.structured_frame(100 * seq + i, *FRAME_HW)
            Image.fromarray(rgb).save(os.path.join(sdir, "color", "%06d.png" % (i + 1)))
            Image.fromarray(depth).save(os.path.join(sdir, "depth_filled", "%06d.png" % (i + 1)))
            for c in classes:
                P = gt_pose(seq, i) if c == CLASS_ID else Fx.pose(1000 * seq + i, (0.05, 0.0, 0.9))
                np.savetxt(os.path.join(sdir, "pose_gt", str(c), "%06d.txt" % (i + 1)), P)
    lines = keyframe_lines()
    for d in ("image_sets", "YCB_Video_toolbox"):
        os.makedirs(os.path.join(ycb, d), exist_ok=True)
        with open(os.path.join(ycb, d, "keyframe.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
    pdir = os.path.join(ycb, "YCB_Video_toolbox", "results_PoseCNN_RSS2018")
    os.makedirs(pdir, exist_ok=True)
    for idx, line in enumerate(lines):
        seq, fid = int(line[:4]), int(line[5:])
        present = [c for c in (OTHER_CLASS, CLASS_ID) if c == OTHER_CLASS or dict((s, h) for s, _, h in SEQUENCES)[seq]]
        rois = np.zeros((len(present), 6))
        poses = np.zeros((len(present), 7))
        for r, c in enumerate(present):                         # the wanted class is NOT the first row
            rois[r] = [0, c] + list(rng.uniform(10, 200, 4))
            q, t = posecnn_pose(seq, fid, c)
            poses[r] = np.r_[q, t]
        savemat(os.path.join(pdir, "%06d.mat" % idx), {"rois": rois, "poses_icp": poses})
    return ycb
