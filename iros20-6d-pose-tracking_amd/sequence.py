"""Headless sequence drivers over the reference's dataset layouts: the loops of predict.py without the GUI.

YCB-Video (predict.py:446-575 `predictSequenceYcb`, :299-443 `getResultsYcb`, :88-123 `use_posecnn_res`):
    <seq_dir>/color/*.png            RGB uint8
    <seq_dir>/depth_filled/*.png     uint16 millimetres
    <seq_dir>/pose_gt/<class_id>/*.txt   4x4 object-in-camera, np.savetxt format
    <ycb_dir>/image_sets/keyframe.txt, <ycb_dir>/YCB_Video_toolbox/results_PoseCNN_RSS2018/%06d.mat   (PoseCNN initialisation)
    <ycb_dir>/YCB_Video_toolbox/PoseRBPF_Results/YCB_results_RGBD/<class folder>/seq_<k>/Pose*.txt    (PoseRBPF initialisation)

predict_sequence_ycb writes `<out_dir>/%05d.txt` (prediction) and `%05dgt.txt` exactly as predict.py:567-569 and returns
the per-frame ADD / ADD-S errors, their AUCs (x100, eval_ycb.py) and the tracking rate.  Checked against the UNMODIFIED
reference drivers run on a synthetic tree (tests/golden/driver_ycbv.npz, tests/test_ycbv_drivers.py)."""
import glob
import os
import time

import numpy as np
from PIL import Image

from . import metrics
from . import utils as U


def read_rgb(path):
    """np.array(Image.open(path)) of predict.py:532 for the 3-channel PNGs of both datasets; a palette / RGBA / grey file is
    converted to RGB here instead of reaching the kernels with another channel count."""
    return np.array(Image.open(path).convert("RGB"))


def read_depth_mm(path):
    d = np.array(Image.open(path))
    return d.astype(np.uint16)


def _keyframes(ycb_dir):
    with open(os.path.join(ycb_dir, "image_sets", "keyframe.txt")) as ff:
        return [line.rstrip() for line in ff]


def _posecnn_pose(ycb_dir, class_id, index):
    """poses_icp row of `class_id` in results_PoseCNN_RSS2018/%06d.mat -> 4x4 (predict.py:114-122 / :366-374)"""
    import scipy.io
    res = scipy.io.loadmat(os.path.join(ycb_dir, "YCB_Video_toolbox", "results_PoseCNN_RSS2018", "%06d.mat" % index))
    row = np.where(res["rois"][:, 1] == class_id)
    tmp = res["poses_icp"][row].reshape(-1)
    if tmp.size < 7:
        raise ValueError("PoseCNN result %06d.mat has no detection of class %d" % (index, class_id))
    pose = np.eye(4)
    pose[:3, :3] = U.quaternion_matrix(tmp[:4])[:3, :3]
    pose[:3, 3] = tmp[4:7]
    return pose


def use_posecnn_res(ycb_dir, class_id, seq_frame_str):
    """predict.py:88-123: the PoseCNN (+ICP) estimate of the keyframe NEAREST to 'SSSS/FFFFFF' -- the frame number is searched
    outwards (n, then n+1 / n-1, n+2 / n-2 ...; the later frame wins a tie) in <ycb_dir>/image_sets/keyframe.txt, and the result
    file is indexed by the keyframe's LINE NUMBER.  (The reference opens the keyframe list through an undefined global and the
    result file through an absolute '/YCB_Video_toolbox/...' path, :90 / :114; both are read under ycb_dir here.)"""
    seq_frames = _keyframes(ycb_dir)
    seq_id, start_frame = int(seq_frame_str.split("/")[0]), int(seq_frame_str.split("/")[1])
    in_seq = [f for f in seq_frames if f.startswith("%04d/" % seq_id)]
    if not in_seq:
        raise ValueError("no keyframe of sequence %04d in keyframe.txt (the reference loops forever here)" % seq_id)
    neighbor = 0
    while True:
        tmp = "%04d/%06d" % (seq_id, start_frame + neighbor)
        if tmp in seq_frames:
            break
        tmp = "%04d/%06d" % (seq_id, start_frame - neighbor)
        if tmp in seq_frames:
            break
        neighbor += 1
    return _posecnn_pose(ycb_dir, class_id, seq_frames.index(tmp))


def find_class_videos_ycb(ycb_dir, class_id, testset=True):
    """Utils.py:108-123 `findClassContainedVideosYcb`: ids of the videos under data_organized/ whose pose_gt/ has the class
    (test set = 48..59)."""
    out = []
    for gt_dir in sorted(glob.glob(os.path.join(ycb_dir, "data_organized", "*", "pose_gt"))):
        vid = int(os.path.basename(os.path.dirname(gt_dir)))
        if testset and (vid < 48 or vid > 59):
            continue
        if class_id in [int(d) for d in os.listdir(gt_dir)]:
            out.append(vid)
    return out


def poserbpf_pose(ycb_dir, class_id, seq_id):
    """predict.py:375-390 / :500-516: first line of PoseRBPF_Results/YCB_results_RGBD/<class_id-th folder>/seq_<k>/Pose*.txt,
    k = 1-based rank of seq_id among the test videos that contain the class; fields 2.. = translation, quaternion (w,x,y,z)."""
    seqs = sorted(find_class_videos_ycb(ycb_dir, class_id, testset=True))
    res_dir = os.path.join(ycb_dir, "YCB_Video_toolbox", "PoseRBPF_Results", "YCB_results_RGBD")
    folders = sorted(os.listdir(res_dir))
    cur = os.path.join(res_dir, folders[class_id - 1], "seq_%d" % (seqs.index(seq_id) + 1))
    with open(glob.glob(os.path.join(cur, "Pose*.txt"))[0]) as ff:
        pose = ff.readlines()[0].rstrip().split()[2:]
    out = np.eye(4)
    out[:3, 3] = [float(v) for v in pose[:3]]
    out[:3, :3] = U.quaternion_matrix([float(v) for v in pose[3:7]])[:3, :3]
    return out


def predict_sequence_ycb(tracker, seq_dir, class_id, out_dir, start_frame=0, reinit=None, max_frames=None, init="gt",
                         reinit_frames=None, ycb_dir=None, seq_id=None):
    """tracker: se3tracknet_amd.Tracker.
    init: 'gt' (what predict.py:447 hard-codes) | 'posecnn' (:480-496: start at the keyframe nearest to start_frame) | 'poserbpf'.
    reinit_frames: the reference's --reinit_frames, a comma-separated string or list of 'SSSS/FFFFFF': before tracking the image
    with index i (0-based; its YCB frame id is i + 1), if 'SSSS/%06d' % (i + 1) is listed, the pose fed in is the PoseCNN estimate
    nearest to frame NUMBER i - 1 (predict.py:538-541, use_posecnn_res).  Needs ycb_dir (the dataset root) and seq_id (default:
    the directory name).  reinit: alternatively {frame_index: 4x4 pose}."""
    rgb_files = sorted(glob.glob(os.path.join(seq_dir, "color", "*")))
    depth_files = sorted(glob.glob(os.path.join(seq_dir, "depth_filled", "*")))
    gt_files = sorted(glob.glob(os.path.join(seq_dir, "pose_gt", str(class_id), "*")))
    assert len(rgb_files) == len(depth_files) == len(gt_files) > start_frame, "incomplete sequence directory"
    gt_poses = [np.loadtxt(f) for f in gt_files]
    if isinstance(reinit_frames, str):
        reinit_frames = reinit_frames.split(",")
    reinit_frames = list(reinit_frames or [])
    if seq_id is None and (reinit_frames or init != "gt"):
        seq_id = int(os.path.basename(os.path.normpath(seq_dir)))
    if (reinit_frames or init != "gt") and ycb_dir is None:
        raise ValueError("PoseCNN / PoseRBPF initialisation needs ycb_dir (image_sets/, YCB_Video_toolbox/)")
    if init == "gt":
        prev_pose = gt_poses[start_frame].copy()          # predict.py:478-479
    elif init == "posecnn":                                # predict.py:480-496
        seq_frames = _keyframes(ycb_dir)
        neighbor = 0
        while True:
            s_ = "%04d/%06d" % (seq_id, start_frame + neighbor)
            if s_ in seq_frames:
                start_frame = start_frame + neighbor
                break
            s_ = "%04d/%06d" % (seq_id, start_frame - neighbor)
            if s_ in seq_frames:
                start_frame = start_frame - neighbor
                break
            neighbor += 1
            if neighbor > 10 ** 6:
                raise ValueError("no keyframe of sequence %04d" % seq_id)
        prev_pose = use_posecnn_res(ycb_dir, class_id, s_)
    elif init == "poserbpf":
        prev_pose = poserbpf_pose(ycb_dir, class_id, seq_id)
    else:
        raise ValueError("init must be 'gt', 'posecnn' or 'poserbpf'")
    end = len(rgb_files) if max_frames is None else min(len(rgb_files), start_frame + 1 + max_frames)
    pred_poses = [prev_pose]
    os.makedirs(out_dir, exist_ok=True)
    t_track = 0.0
    for i in range(start_frame + 1, end):
        rgb = read_rgb(rgb_files[i])
        depth = read_depth_mm(depth_files[i])
        A_in_cam = prev_pose.copy()
        if reinit and i in reinit:
            A_in_cam = np.asarray(reinit[i], np.float64).copy()
        if reinit_frames and "%04d/%06d" % (seq_id, i + 1) in reinit_frames:          # predict.py:538-541
            A_in_cam = use_posecnn_res(ycb_dir, class_id, "%04d/%06d" % (seq_id, i - 1))
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
        res["add_auc"] = metrics.auc(add_errs)    # 0.0 when no frame is below 0.1 m (the reference's VOCap raises)
        res["adi_auc"] = metrics.auc(adi_errs)
    return res


YCB_TEST_SEQUENCES = tuple(range(48, 60))   # predict.py:349 skips every video outside 0048..0059


def get_results_ycb(tracker, ycb_dir, class_id, out_dir, seq_ids=YCB_TEST_SEQUENCES, max_frames=None, initialize_method="gt",
                    lockstep=False):
    """The loop of predict.py:299-443 `getResultsYcb` (GT initialisation, no re-init, no video): every TEST
    sequence (48..59 by default, as predict.py:349; seq_ids=None takes every directory) under
    <ycb_dir>/data_organized/ that has pose_gt/<class_id>/ is tracked from its first frame and written as
    <out_dir>/seq<ID>/%07d.txt -- the layout eval_one_class / the reference's eval_ycb.py:95-96 parse
    (file index = frame id - 1).  initialize_method: 'gt' (what predict.py:301 hard-codes) | 'posecnn' (:362-373: the PoseCNN
    result of the keyframe 'SSSS/000001') | 'poserbpf' (:374-390): two of the three result columns the reference publishes
    start from those.  Returns {seq_id: n_poses}.
    lockstep = True (extension): the sequences of the class advance TOGETHER, frame index by frame index, through
    ``tracker.on_track_batch`` (one se3tn_on_track_batch call per step for all sequences still running, in chunks of the tracker's
    max_samples) instead of one after the other -- frames of one sequence stay serial, the sequences are independent; per sequence
    the same arithmetic (up to 5 running sequences: the same bits as the serial loop).  The files written are the same."""
    if initialize_method not in ("gt", "posecnn", "poserbpf"):
        raise ValueError("initialize_method must be 'gt', 'posecnn' or 'poserbpf'")
    root = os.path.join(ycb_dir, "data_organized")
    seqs = []
    for seq_dir in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(os.path.join(seq_dir, "pose_gt", str(class_id))):
            continue
        seq_id = int(os.path.basename(seq_dir))
        if seq_ids is not None and seq_id not in seq_ids:
            continue
        rgb_files = sorted(glob.glob(os.path.join(seq_dir, "color", "*")))
        depth_files = sorted(glob.glob(os.path.join(seq_dir, "depth_filled", "*")))
        gt_files = sorted(glob.glob(os.path.join(seq_dir, "pose_gt", str(class_id), "*")))
        assert len(rgb_files) == len(depth_files) == len(gt_files) > 0, "incomplete sequence directory %s" % seq_dir
        n = len(rgb_files) if max_frames is None else min(len(rgb_files), max_frames)
        if initialize_method == "posecnn":
            prev_pose = _posecnn_pose(ycb_dir, class_id, _keyframes(ycb_dir).index("%04d/%06d" % (seq_id, 1)))
        elif initialize_method == "poserbpf":
            prev_pose = poserbpf_pose(ycb_dir, class_id, seq_id)
        else:
            prev_pose = np.loadtxt(gt_files[0])
        seqs.append(dict(id=seq_id, rgb=rgb_files, depth=depth_files, n=n, prev=prev_pose, pred=[prev_pose]))
    if lockstep:
        cap = int(tracker.engine.max_batch)
        for i in range(1, max([s["n"] for s in seqs] or [0])):
            running = [s for s in seqs if i < s["n"]]
            for c0 in range(0, len(running), cap):
                chunk = running[c0:c0 + cap]
                poses = tracker.on_track_batch([s["prev"] for s in chunk], [read_rgb(s["rgb"][i]) for s in chunk],
                                               [read_depth_mm(s["depth"][i]) for s in chunk])
                for s, p in zip(chunk, poses):
                    s["prev"] = np.array(p, np.float64)
                    s["pred"].append(s["prev"])
    else:
        for s in seqs:
            for i in range(1, s["n"]):
                cur = tracker.on_track(s["prev"], read_rgb(s["rgb"][i]), read_depth_mm(s["depth"][i]))
                s["prev"] = cur.copy()
                s["pred"].append(cur)
    done = {}
    for s in seqs:
        sdir = os.path.join(out_dir, "seq%d" % s["id"])
        os.makedirs(sdir, exist_ok=True)
        for i, p in enumerate(s["pred"]):
            np.savetxt(os.path.join(sdir, "%07d.txt" % i), p)
        done[s["id"]] = len(s["pred"])
    return done


def eval_one_class(res_dir, ycb_dir, class_id):
    """eval_ycb.py:67-119: ADD / ADD-S AUC (x100) of the keyframe poses found under res_dir/seq*/,
    against <ycb_dir>/data_organized/%04d/pose_gt/<class_id>/%06d.txt, with the class's
    CADmodels/*/points.xyz as the model and YCB_Video_toolbox/keyframe.txt as the frame filter."""
    pose_files = sorted(glob.glob(os.path.join(res_dir, "**", "*.txt"), recursive=True))
    assert len(pose_files) > 0, "no pose files under %s" % res_dir
    model_files = sorted(glob.glob(os.path.join(ycb_dir, "CADmodels", "**", "points.xyz"), recursive=True))
    model_pts = np.loadtxt(model_files[class_id - 1]).reshape(-1, 3)
    with open(os.path.join(ycb_dir, "YCB_Video_toolbox", "keyframe.txt")) as ff:
        keyframes = set(line.rstrip() for line in ff)
    adi_errs, add_errs = [], []
    for pose_file in pose_files:
        rel = os.path.relpath(pose_file, res_dir).split(os.sep)
        seq_id = int(rel[0].replace("seq", ""))
        frame_id = int(os.path.basename(pose_file).split(".")[0]) + 1
        if "%04d/%06d" % (seq_id, frame_id) not in keyframes:
            continue
        pred = np.loadtxt(pose_file)
        gt = np.loadtxt(os.path.join(ycb_dir, "data_organized", "%04d" % seq_id, "pose_gt", str(class_id), "%06d.txt" % frame_id))
        adi_errs.append(metrics.adi(pred, gt, model_pts))
        add_errs.append(metrics.add(pred, gt, model_pts))
    assert len(adi_errs) > 0, "no keyframe among the result files"
    adi_errs = np.sort(np.array(adi_errs)); add_errs = np.sort(np.array(add_errs))
    return {"add_auc": metrics.auc(add_errs), "adi_auc": metrics.auc(adi_errs),
            "adi_errs": adi_errs, "add_errs": add_errs, "n": len(adi_errs)}


YCBINEOAT_OBJECTS = ("cracker", "bleach", "sugar", "tomato", "mustard")   # eval_ycbineoat.py:48


def predict_sequence_ycbineoat(tracker, data_dir, out_dir, max_frames=None):
    """predict.py:578-626 `predictSequenceYcbInEOAT` without the GUI: <data_dir>/rgb/*.png,
    depth_filled/*.png (uint16 mm), annotated_poses/*.txt; the track starts from the first annotated
    pose and -- unlike the YCB-Video driver -- frame 0 itself is tracked too; every pose is written as
    <out_dir>/%07d.txt.  The reference builds this Tracker with rot_normalizer = 30 deg (:586); pass a
    tracker constructed that way.  Returns {"poses": [n,4,4], "hz": ...}."""
    rgb_files = sorted(glob.glob(os.path.join(data_dir, "rgb", "*.png")))
    depth_files = sorted(glob.glob(os.path.join(data_dir, "depth_filled", "*.png")))
    gt_files = sorted(glob.glob(os.path.join(data_dir, "annotated_poses", "*.txt")))
    assert len(rgb_files) == len(depth_files) > 0 and len(gt_files) > 0, "incomplete YCBInEOAT directory"
    n = len(rgb_files) if max_frames is None else min(len(rgb_files), max_frames)
    prev_pose = np.loadtxt(gt_files[0]).copy()
    os.makedirs(out_dir, exist_ok=True)
    poses, t_track = [], 0.0
    for i in range(n):
        rgb = read_rgb(rgb_files[i])
        depth = read_depth_mm(depth_files[i])
        t0 = time.perf_counter()
        cur_pose = tracker.on_track(prev_pose.copy(), rgb, depth, gt_A_in_cam=np.eye(4), gt_B_in_cam=np.eye(4))
        t_track += time.perf_counter() - t0
        prev_pose = cur_pose.copy()
        np.savetxt(os.path.join(out_dir, "%07d.txt" % i), cur_pose)
        poses.append(cur_pose)
    return {"poses": np.array(poses), "frames": n, "hz": n / t_track if t_track > 0 else float("nan")}


def eval_ycbineoat(res_dir, data_dir, ycb_dir, objects=YCBINEOAT_OBJECTS):
    """eval_ycbineoat.py:45-109 `eval_all`: every folder of res_dir (one per video, named after it) is
    matched to an object by substring, its %07d.txt poses are compared one-to-one with
    <data_dir>/<folder>/annotated_poses/*.txt, the model is the CADmodels/*/points.xyz whose path
    contains the object name.  Returns per-object and overall ADD / ADD-S AUC (x100)."""
    models = {}
    for t in sorted(glob.glob(os.path.join(ycb_dir, "CADmodels", "*", "points.xyz"))):
        for obj in objects:
            if obj in t:
                models[obj] = np.loadtxt(t).reshape(-1, 3)
    class_res = {obj: {"add": [], "add-s": []} for obj in objects}
    for folder in sorted(os.listdir(res_dir)):
        if ".tar.gz" in folder or not os.path.isdir(os.path.join(res_dir, folder)):
            continue
        obj = next((o for o in objects if o in folder), None)
        assert obj is not None, "result folder %s names no known object" % folder
        pred_files = sorted(glob.glob(os.path.join(res_dir, folder, "*.txt")))
        gt_files = sorted(glob.glob(os.path.join(data_dir, folder, "annotated_poses", "*.txt")))
        assert len(pred_files) == len(gt_files), "#pred_files:%d, #gt_files:%d" % (len(pred_files), len(gt_files))
        for pf, gf in zip(pred_files, gt_files):
            pred, gt = np.loadtxt(pf), np.loadtxt(gf)
            class_res[obj]["add"].append(metrics.add(pred, gt, models[obj]))
            class_res[obj]["add-s"].append(metrics.adi(pred, gt, models[obj]))

    auc = metrics.auc   # 0.0 when nothing is below 0.1 m (the reference's VOCap raises) or the list is empty
    out = {"per_object": {}, "n": 0}
    adds, adis = [], []
    for obj in objects:
        a, s_ = class_res[obj]["add"], class_res[obj]["add-s"]
        if not a:
            continue
        out["per_object"][obj] = {"add_auc": auc(a), "adi_auc": auc(s_), "n": len(a)}
        adds += a; adis += s_
    out.update(add_auc=auc(adds), adi_auc=auc(adis), n=len(adis))
    return out


def eval_all_classes(per_class):
    """eval_ycb.py:121-161 `eval_all`: per_class = {class_id: result of eval_one_class}; the overall ADD /
    ADD-S AUC is VOCap over the CONCATENATED keyframe errors of all classes (not the mean of the AUCs)."""
    adi = np.concatenate([per_class[k]["adi_errs"] for k in sorted(per_class)])
    add = np.concatenate([per_class[k]["add_errs"] for k in sorted(per_class)])
    return {"add_auc": metrics.auc(add), "adi_auc": metrics.auc(adi), "n": len(adi),
            "per_class": {k: {"add_auc": per_class[k]["add_auc"], "adi_auc": per_class[k]["adi_auc"], "n": per_class[k]["n"]}
                          for k in sorted(per_class)}}


def eval_objects_parallel(class_ids, run_class, rank=0, world=1, group=None):
    """BASELINE configs[4]: every rank (one per GPU) tracks and evaluates the classes it owns
    (dist.shard_round_robin), `run_class(class_id)` -> result of eval_one_class (typically: build that
    object's Tracker from its own checkpoint, get_results_ycb, eval_one_class); the per-class results are
    all-gathered (host objects) and every rank returns the same eval_all_classes aggregate."""
    from . import dist as D
    mine, failure = {}, None
    for cid in D.shard_round_robin(list(class_ids), rank, world):
        try:
            mine[int(cid)] = run_class(int(cid))
        except Exception as e:   # noqa: BLE001 -- every rank must still enter the collective below
            failure = "rank %d, class %d: %r" % (rank, int(cid), e)
            break
    if world > 1:
        merged, failures = {}, []
        for part, fail in D.gather_objects((mine, failure), group):
            merged.update(part)
            if fail:
                failures.append(fail)
    else:
        merged, failures = mine, [failure] if failure else []
    if failures:   # raised on EVERY rank, after the gather: nobody is left waiting in a collective
        raise RuntimeError("eval_objects_parallel: " + "; ".join(failures))
    assert sorted(merged) == sorted(int(c) for c in class_ids), "a class was not evaluated"
    return eval_all_classes(merged)
