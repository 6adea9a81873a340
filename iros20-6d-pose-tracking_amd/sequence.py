"""Headless sequence driver over the reference's YCB-Video directory layout:
the loop of predict.py:446-575 `predictSequenceYcb` without the GUI.

    <seq_dir>/color/*.png            RGB uint8
    <seq_dir>/depth_filled/*.png     uint16 millimetres
    <seq_dir>/pose_gt/<class_id>/*.txt   4x4 object-in-camera, np.savetxt format

Writes `<out_dir>/%05d.txt` (prediction) and `%05dgt.txt` exactly as predict.py:567-569 and returns
the per-frame ADD / ADD-S errors, their AUCs (x100, eval_ycb.py) and the tracking rate."""
import glob
import os
import time

import numpy as np
from PIL import Image

from . import metrics


def read_rgb(path):
    return np.array(Image.open(path).convert("RGB"))


def read_depth_mm(path):
    d = np.array(Image.open(path))
    return d.astype(np.uint16)


def predict_sequence_ycb(tracker, seq_dir, class_id, out_dir, start_frame=0, reinit=None, max_frames=None):
    """tracker: se3tracknet_amd.Tracker; reinit: optional {frame_index: 4x4 pose} (the reference
    re-initialises from PoseCNN at listed frames, predict.py:539-541)."""
    rgb_files = sorted(glob.glob(os.path.join(seq_dir, "color", "*")))
    depth_files = sorted(glob.glob(os.path.join(seq_dir, "depth_filled", "*")))
    gt_files = sorted(glob.glob(os.path.join(seq_dir, "pose_gt", str(class_id), "*")))
    assert len(rgb_files) == len(depth_files) == len(gt_files) > start_frame, "incomplete sequence directory"
    gt_poses = [np.loadtxt(f) for f in gt_files]
    end = len(rgb_files) if max_frames is None else min(len(rgb_files), start_frame + 1 + max_frames)
    prev_pose = gt_poses[start_frame].copy()          # init == 'gt' (predict.py:478-479)
    pred_poses = [prev_pose]
    os.makedirs(out_dir, exist_ok=True)
    t_track = 0.0
    for i in range(start_frame + 1, end):
        rgb = read_rgb(rgb_files[i])
        depth = read_depth_mm(depth_files[i])
        A_in_cam = prev_pose.copy()
        if reinit and i in reinit:
            A_in_cam = np.asarray(reinit[i], np.float64).copy()
        t0 = time.perf_counter()
        cur_pose = tracker.on_track(A_in_cam, rgb, depth, gt_A_in_cam=gt_poses[i - 1], gt_B_in_cam=gt_poses[i])
        t_track += time.perf_counter() - t0
        prev_pose = cur_pose.copy()
        pred_poses.append(cur_pose)
    pred_poses = np.array(pred_poses)
    add_errs, adi_errs = [], []
    for k in range(len(pred_poses)):
        np.savetxt(os.path.join(out_dir, "%05d.txt" % k), pred_poses[k])
        np.savetxt(os.path.join(out_dir, "%05dgt.txt" % k), gt_poses[start_frame + k])
        if tracker.object_cloud is not None:
            add_errs.append(metrics.add(pred_poses[k], gt_poses[start_frame + k], tracker.object_cloud))
            adi_errs.append(metrics.adi(pred_poses[k], gt_poses[start_frame + k], tracker.object_cloud))
    res = {"poses": pred_poses, "frames": len(pred_poses) - 1,
           "hz": (len(pred_poses) - 1) / t_track if t_track > 0 else float("nan")}
    if adi_errs:
        res.update(add_errs=np.array(add_errs), adi_errs=np.array(adi_errs))
        for name, errs in (("add_auc", add_errs), ("adi_auc", adi_errs)):
            try:
                res[name] = metrics.VOCap(np.array(errs)) * 100
            except IndexError:  # no frame below 0.1 m: the reference's VOCap raises
                res[name] = 0.0
    return res
