"""Free-running two-track comparison (the reachable half of BASELINE configs[2]'s
"bit-identical pose track" and SURVEY.md section 7's "identical integer bbox sequence and poses within 1e-5 per frame").

TEST INFRASTRUCTURE ONLY (imports the oracle).  Used by tests/test_free_run.py, bench.py's `track.free_running`
block and scripts/free_run_report.py; never by the product package.

oracle/closed_loop.py:run_regime is PER-STEP parity: every frame the oracle is evaluated at the pose the HIP tracker is
at, so errors cannot compound.  Here two closed loops run INDEPENDENTLY from the same start over the same camera frames
(the loop of predict.py:416-420 / :529-564, prev_pose <- cur_pose):

    HIP track      P^h_{f+1} = next(P^h_f, Tracker.on_track(P^h_f, rgb_f, depth_f))      (renders its own image A on the GPU)
    oracle track   P^o_{f+1} = next(P^o_f, O.on_track(sd, P^o_f, rgb_f, depth_f, A^o_f))  (A^o_f = oracle/ss_fast.py render at P^o_f)

on two PROBLEMS (classes below), plus a CONTROL track per problem -- the oracle with channels-last network inputs: the same torch-CPU
arithmetic in another summation order, i.e. how far the reference path separates from ITSELF:
  * RandomInitProblem: the stand-in of oracle/closed_loop.py -- random-init weights with calibrated FC biases, frames that do not
    contain the object, `next` = closed_loop._next_pose (rotation fully fed back, the network's own translation step carried on a
    seeded anchor trajectory; a random-init network has no reason to stay on the object);
  * SynthTrackProblem: oracle/synth_track.py -- an object that IS in the frames, ground-truth poses, stand-in weights TRAINED on it
    (tests/golden/synth_tracker*.npz, one per normaliser regime), `next` = the identity: predict.py:416-420 unmodified.
Nothing of one side enters the other.  The oracle tracks run in worker processes (they cost ~0.06-0.1 s of CPU per frame; the HIP
tracks 0.2 ms), which get the HIP track's images only to COUNT differing pixels.

What is reported per (regime, seed) track pair, over `frames` frames:
  * first frame whose integer bbox differs, number of frames with a differing bbox (the only discrete decisions on the
    path besides the rasteriser's coverage);
  * first frame whose image A differs, number of such frames, differing pixels per frame (median / max over the differing frames);
  * separation of the two tracks over time: max |dP| over the 4x4 (the north-star's pose measure), rotation angle between the
    two rotations [deg], translation distance [mm] -- overall, at checkpoints, and per window of 100 frames (does it stay at
    the per-step rounding level, grow like a random walk, or run away?);
  * ADD and ADD-S between the two tracks on the object's model points (product `metrics.py`; Utils.py:72-98), and the
    ADD-S AUC the HIP track would score if the oracle track were the ground truth (eval_ycb.py:45-64; 100 = identical).

A random-init network is NOT a tracker: nothing pulls a perturbed pose back (a trained se(3)-TrackNet regresses the residual
to the observed frame, so a perturbation of the pose is corrected by the next frame's estimate).  The random-init figures are
therefore an upper bound on what rounding differences can do to a track of this length -- the open-loop sensitivity of the
pose -> image A -> network -> pose map with no restoring force -- and the trained-weights figures are the ones that say what
"the same track" means (measured: DESIGN.md section 4, profiles/r06_free_run.json)."""
import os
import tempfile
import time

import numpy as np

from . import closed_loop as CL
from . import fixtures as Fx
from . import se3_oracle as O

CHECKPOINTS = (1, 3, 10, 30, 100, 300, 1000, 3000)
WINDOW = 100


def track_setup(seed):
    """What distinguishes the seeds: the weights (O.make_state_dict(seed)), the start rotation, the phase of the anchor
    trajectory and the offset into the frame cycle."""
    return dict(weights_seed=seed, pose_seed=3 + 11 * seed, phase=37 * seed, frame_offset=5 * seed)


def initial_pose(seed):
    s = track_setup(seed)
    P = Fx.pose(s["pose_seed"], (0.0, 0.0, 0.8))
    P[:3, 3] = CL.anchor(s["phase"])
    return P


def next_pose(P, Q, f, phase):
    """closed_loop._next_pose with a per-seed phase of the anchor trajectory"""
    N = Q.copy()
    N[:3, 3] = CL.anchor(f + 1 + phase) + (Q[:3, 3] - P[:3, 3])
    if CL._lost(N):
        N[:3, 3] = CL.anchor(f + 1 + phase)
        return N, 1
    return N, 0


def calibrated_weights(seed, mesh, seq, K, numpy_rule="numpy1"):
    """closed_loop.make_tracker's calibration, oracle side only: FC biases re-centred on N_CALIB (pose, frame) pairs whose
    image A the oracle renders itself.  The logits do not depend on the normalisers: one state_dict serves both regimes."""
    import torch
    mean, std = Fx.mean_std(0)
    sd = O.make_state_dict(seed, head_gain=CL.HEAD_GAIN)
    om = CL.oracle_mesh(mesh)
    logits = []
    for i in range(CL.N_CALIB):
        P = Fx.pose(100 + i, tuple(CL.anchor(25 * i)))
        rgbA, depthA = CL.oracle_image_A(om, P, K, CL.OBJECT_WIDTH_MM, numpy_rule)
        rgb, depth = seq[i % CL.N_DISTINCT_FRAMES]
        _, aux = O.on_track(sd, P, rgb, depth, rgbA, depthA, K, CL.OBJECT_WIDTH_MM, mean, std)
        logits.append(np.r_[aux["trans_logit"], aux["rot_logit"]])
    centre = np.mean(logits, 0).astype(np.float32)
    sd["trans_out.0.bias"] = sd["trans_out.0.bias"] - torch.from_numpy(centre[:3])
    sd["rot_out.0.bias"] = sd["rot_out.0.bias"] - torch.from_numpy(centre[3:])
    return sd


class RandomInitProblem:
    """oracle/closed_loop.py's stand-in: random-init weights with calibrated FC biases, a random-colour sphere, structured frames that
    do NOT contain the object, rotation fed back, translation step carried on the anchor trajectory."""
    kind = "random_init"

    def __init__(self, seed, regime, sd=None, mesh=None, subdiv=5):
        from . import raster_oracle as R
        self.seed, self.regime = seed, regime
        self.mesh = mesh if mesh is not None else R.icosphere(subdiv, 0.06, 0)
        self.K = camera_matrix()
        self.mean, self.std = Fx.mean_std(0)
        self.tn, self.rn = CL.REGIMES[regime]
        self.object_width = CL.OBJECT_WIDTH_MM
        self.seq = CL._frames()
        self.sd = sd if sd is not None else calibrated_weights(seed, self.mesh, self.seq, self.K)
        self.s = track_setup(seed)

    def spec(self):
        return dict(kind=self.kind, seed=self.seed, regime=self.regime, sd={k: v.numpy() for k, v in self.sd.items()},
                    mesh={k: np.asarray(v) for k, v in self.mesh.items()})

    def frame(self, f):
        return self.seq[(f + self.s["frame_offset"]) % CL.N_DISTINCT_FRAMES]

    def initial_pose(self):
        return initial_pose(self.seed)

    def next_pose(self, P, Q, f):
        return next_pose(P, Q, f, self.s["phase"])

    def ground_truth(self, f):
        return None


class SynthTrackProblem:
    """oracle/synth_track.py: an object that IS in the frames, ground-truth poses, weights trained on it (tests/golden/
    synth_tracker.npz) -- the loop of predict.py:416-420 exactly: prev_pose <- on_track(prev_pose, frame), started from the
    ground-truth pose of frame 0 (predict.py:404-409), no anchor."""
    kind = "synth"

    def __init__(self, seed, sequence, weights=None, regime=None):
        from . import synth_track as ST
        self.seed, self.regime = seed, regime or ST.REGIME
        self.mesh = ST.make_object()
        self.K = camera_matrix()
        self.tn, self.rn = ST.normalizers(self.regime)
        self.object_width = ST.OBJECT_WIDTH_MM
        self.sequence_path = sequence if isinstance(sequence, str) else None
        self.sequence = ST.Sequence.load(sequence) if isinstance(sequence, str) else sequence
        self.weights_path = weights or default_synth_weights(self.regime)
        self.sd, self.mean, self.std, self.weights_info = load_synth_weights(self.weights_path)
        self.gt = [ST.gt_pose(seed, f, self.regime) for f in range(len(self.sequence))]

    def spec(self):
        assert self.sequence_path, "save the sequence first (Sequence.save) so that the workers can load it"
        return dict(kind=self.kind, seed=self.seed, sequence=self.sequence_path, weights=self.weights_path, regime=self.regime)

    def frame(self, f):
        return self.sequence.frame(f + 1)                 # iteration f estimates the pose of camera frame f + 1 from that of frame f

    def initial_pose(self):
        return self.gt[0].copy()

    def next_pose(self, P, Q, f):
        G = self.gt[f + 1]
        if np.linalg.norm(Q[:3, 3] - G[:3, 3]) > 0.05 or not np.isfinite(Q).all():     # lost: re-detected at the ground truth (counted)
            return G.copy(), 1
        return Q.copy(), 0

    def ground_truth(self, f):
        return self.gt[f]


def default_synth_weights(regime=None):
    """the trained stand-in of a normaliser regime: synth_tracker.npz (30 degrees, predict.py:586), synth_tracker_5deg.npz (predict.py:128)"""
    name = "synth_tracker_5deg.npz" if regime == "ycb_video_5deg" else "synth_tracker.npz"
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name)


def load_synth_weights(path):
    """(state_dict, mean [8], std [8], info) of the trained stand-in: O.make_state_dict(base_seed) with the trained subset (stored
    as float16 values) written over it -- scripts/train_synth_tracker.py"""
    import torch
    z = np.load(path)
    sd = O.make_state_dict(int(z["base_seed"]))
    n = 0
    for k in z.files:
        if k.startswith("w:"):
            assert sd[k[2:]].shape == z[k].shape, k
            sd[k[2:]] = torch.from_numpy(z[k].astype(np.float32))
            n += 1
    info = dict(trained_tensors=n, held_out=[float(v) for v in z["val"]] if "val" in z.files else None)
    return sd, np.asarray(z["mean"], np.float64), np.asarray(z["std"], np.float64), info


def problem_from_spec(spec):
    import torch
    if spec["kind"] == "random_init":
        sd = {k: torch.from_numpy(np.array(v)) for k, v in spec["sd"].items()}
        return RandomInitProblem(spec["seed"], spec["regime"], sd=sd, mesh=spec["mesh"])
    return SynthTrackProblem(spec["seed"], spec["sequence"], spec["weights"], spec.get("regime"))


def oracle_track(job):
    """Worker (spawned process, CPU only): the oracle's OWN closed loop on job["problem"] (a Problem.spec()).  Other keys: frames,
    numpy_rule, threads, images (path of the other track's image A stack, or None).  Returns poses fed in [F+1,4,4], outputs
    [F,6], bboxes [F,8], per-frame differing image-A pixels vs the other track, reinits, seconds.
    `variant` = "channels_last" / `save_to`: the CONTROL -- the same reference arithmetic (torch-CPU) with the network's inputs in
    the other memory format, i.e. another summation order inside the convolutions (1.2e-6 on the outputs); its
    images are written out for the other side's pixel count."""
    import torch
    torch.set_num_threads(int(job.get("threads", 2)))
    t0 = time.time()
    pb = problem_from_spec(job["problem"])
    sd, K, mean, std, tn, rn, width = pb.sd, pb.K, pb.mean, pb.std, pb.tn, pb.rn, pb.object_width
    om = CL.oracle_mesh(pb.mesh)
    other = None
    if job.get("images"):
        other = (np.load(job["images"] + ".rgb.npy", mmap_mode="r"), np.load(job["images"] + ".depth.npy", mmap_mode="r"))
    F = int(job["frames"])
    P = pb.initial_pose()
    poses, outs, bboxes, px, reinits = [P.copy()], [], [], [], 0
    keep_rgb, keep_depth = [], []
    variant = job.get("variant")                   # None | "channels_last": the control (the reference path against itself)
    rule = job.get("numpy_rule", "numpy1")
    for f in range(F):
        rgb, depth = pb.frame(f)
        rgbA, depthA = CL.oracle_image_A(om, P, K, width, rule)
        # O.on_track, spelled out so that the control can change the memory format of the network's inputs
        bb = O.compute_bbox(P, K, width, scale=(1000, 1000, 1000))
        rgbB, depthB = O.crop_bbox(rgb, depth, bb, (rgbA.shape[1], rgbA.shape[0]))
        a, b = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std, rule)
        A, B = torch.from_numpy(a)[None], torch.from_numpy(b)[None]
        if variant == "channels_last":
            A, B = A.contiguous(memory_format=torch.channels_last), B.contiguous(memory_format=torch.channels_last)
        o = O.forward(sd, A, B)
        aux = dict(bbox=bb, trans=o["trans"][0].numpy(), rot=o["rot"][0].numpy())
        Q = O.process_predict(P, aux["trans"], aux["rot"], tn, rn)
        if other is not None:
            px.append(int((rgbA != other[0][f]).any(2).sum() + (depthA != other[1][f]).sum()))
        if job.get("save_to"):
            keep_rgb.append(rgbA); keep_depth.append(depthA)
        outs.append(np.r_[aux["trans"], aux["rot"]])
        bboxes.append(np.asarray(aux["bbox"]).reshape(-1))
        P, r = pb.next_pose(P, Q, f)
        reinits += r
        poses.append(P.copy())
    if job.get("save_to"):
        np.save(job["save_to"] + ".rgb.npy", np.array(keep_rgb))
        np.save(job["save_to"] + ".depth.npy", np.array(keep_depth))
    return dict(poses=np.array(poses), outs=np.array(outs), bboxes=np.array(bboxes), px=np.array(px, np.int64),
                reinits=reinits, seconds=time.time() - t0, regime=pb.regime, seed=pb.seed)


def make_hip_tracker(se3, pb, max_samples=1):
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=pb.object_width), pb.mean, pb.std, {"state_dict": pb.sd},
                      trans_normalizer=pb.tn, rot_normalizer=pb.rn, max_samples=max_samples)
    trk.renderer = se3.HipRenderer(trk.engine, pb.mesh)
    assert np.array_equal(trk.K, pb.K)
    return trk


def hip_track(trk, pb, frames, images_path=None):
    """The drop-in Tracker's own closed loop on the problem (one se3tn_on_track call per frame).  Image A of every frame is read
    back (for the pixel comparison only) and written to `images_path`."""
    P = pb.initial_pose()
    poses, outs, bboxes, reinits = [P.copy()], [], [], 0
    rgbs = np.empty((frames, 176, 176, 3), np.uint8) if images_path else None
    deps = np.empty((frames, 176, 176), np.uint16) if images_path else None
    t_on = 0.0
    for f in range(frames):
        rgb, depth = pb.frame(f)
        t0 = time.perf_counter()
        Q = trk.on_track(P, rgb, depth)
        t_on += time.perf_counter() - t0
        lp = trk.last_prediction
        outs.append(np.r_[lp["trans"][0], lp["rot"][0]])
        bboxes.append(np.asarray(lp["bbox"]).reshape(-1))
        if images_path:
            rgbs[f] = trk.renderer.rgb.cpu().numpy()
            deps[f] = trk.renderer.depth.cpu().numpy().view(np.uint16)
        P, r = pb.next_pose(P, Q, f)
        reinits += r
        poses.append(P.copy())
    if images_path:
        np.save(images_path + ".rgb.npy", rgbs)
        np.save(images_path + ".depth.npy", deps)
    return dict(poses=np.array(poses), outs=np.array(outs), bboxes=np.array(bboxes), reinits=reinits,
                ms_per_frame=t_on / max(frames, 1) * 1e3)


def _rot_angle_deg(Ra, Rb):
    c = (np.einsum("nij,nij->n", Ra, Rb) - 1.0) / 2.0
    # small angles from the skew part (acos loses everything below 1e-8 rad)
    D = np.einsum("nij,nkj->nik", Ra, Rb)
    s = 0.5 * np.sqrt((D[:, 2, 1] - D[:, 1, 2]) ** 2 + (D[:, 0, 2] - D[:, 2, 0]) ** 2 + (D[:, 1, 0] - D[:, 0, 1]) ** 2)
    return np.degrees(np.arctan2(s, np.clip(c, -1, 1)))


def compare_tracks(a, b, model_points, metrics=None):
    """a, b: results of hip_track / oracle_track over the same frames; b["px"] = differing image-A pixels per frame.
    `metrics`: the product's metrics module (ADD / ADD-S / VOCap), handed in by the caller (tests / bench)."""
    F = len(a["outs"])
    Pa, Pb = a["poses"], b["poses"]
    sep = np.abs(Pa - Pb).reshape(F + 1, -1).max(1)            # pose fed into frame f (index F: the final estimate)
    ang = _rot_angle_deg(Pa[:, :3, :3], Pb[:, :3, :3])
    dist = np.linalg.norm(Pa[:, :3, 3] - Pb[:, :3, 3], axis=1) * 1e3
    dout = np.abs(a["outs"] - b["outs"]).max(1)
    bbd = (a["bboxes"] != b["bboxes"]).any(1)
    px = np.asarray(b.get("px", []), np.int64)
    out = {"frames": F,
           "first_bbox_divergence_frame": int(np.argmax(bbd)) if bbd.any() else None,
           "bbox_differing_frames": int(bbd.sum()),
           "bbox_max_abs_diff_px": int(np.abs(a["bboxes"].astype(np.int64) - b["bboxes"]).max()),
           "first_pose_divergence_frame": int(np.argmax(sep > 0)) if (sep > 0).any() else None,
           "max_abs_pose_separation": float(sep.max()), "median_abs_pose_separation": float(np.median(sep)),
           "final_abs_pose_separation": float(sep[-1]),
           "max_rotation_separation_deg": float(ang.max()), "final_rotation_separation_deg": float(ang[-1]),
           "max_translation_separation_mm": float(dist.max()), "final_translation_separation_mm": float(dist[-1]),
           "max_abs_trans_rot_output_diff": float(dout.max()), "median_abs_trans_rot_output_diff": float(np.median(dout)),
           "frames_within_1e-5": int((sep[1:] <= 1e-5).sum()), "frames_within_1e-4": int((sep[1:] <= 1e-4).sum()),
           "frames_within_1e-3": int((sep[1:] <= 1e-3).sum()),
           "pose_separation_at_frame": {str(c): float(sep[:c + 1].max()) for c in CHECKPOINTS if c <= F},
           "pose_separation_by_window_of_%d" % WINDOW: [float(sep[w:w + WINDOW].max()) for w in range(1, F + 1, WINDOW)],
           "reinits": [int(a["reinits"]), int(b["reinits"])]}
    if len(px) == F:
        d = px > 0
        out.update(first_imageA_divergence_frame=int(np.argmax(d)) if d.any() else None, imageA_differing_frames=int(d.sum()),
                   imageA_differing_pixels_median=int(np.median(px[d])) if d.any() else 0,
                   imageA_differing_pixels_max=int(px.max()) if F else 0)
        out["imageA_differing_pixels_by_window_of_%d" % WINDOW] = [int(px[w:w + WINDOW].max()) for w in range(0, F, WINDOW)]
    if metrics is not None and model_points is not None:
        add = np.array([metrics.add(Pa[f], Pb[f], model_points) for f in range(1, F + 1)])
        adi = np.array([metrics.adi(Pa[f], Pb[f], model_points, workers=1) for f in range(1, F + 1)])
        out.update(add_between_tracks_mm={"max": float(add.max() * 1e3), "median": float(np.median(add) * 1e3), "final": float(add[-1] * 1e3)},
                   adds_between_tracks_mm={"max": float(adi.max() * 1e3), "median": float(np.median(adi) * 1e3), "final": float(adi[-1] * 1e3)},
                   add_auc_vs_oracle_track=round(metrics.auc(add), 4), adds_auc_vs_oracle_track=round(metrics.auc(adi), 4))
    return out


def summarise(per_track):
    """worst case / totals over the (regime, seed) track pairs of one regime"""
    v = list(per_track.values())
    firsts = [t["first_bbox_divergence_frame"] for t in v if t["first_bbox_divergence_frame"] is not None]
    firsti = [t["first_imageA_divergence_frame"] for t in v if t.get("first_imageA_divergence_frame") is not None]
    out = {"tracks": len(v), "frames_per_track": v[0]["frames"],
           "earliest_bbox_divergence_frame": min(firsts) if firsts else None,
           "bbox_differing_frames": sum(t["bbox_differing_frames"] for t in v),
           "earliest_imageA_divergence_frame": min(firsti) if firsti else None,
           "imageA_differing_frames": sum(t.get("imageA_differing_frames", 0) for t in v),
           "max_abs_pose_separation": max(t["max_abs_pose_separation"] for t in v),
           "median_abs_pose_separation": float(np.median([t["median_abs_pose_separation"] for t in v])),
           "max_rotation_separation_deg": max(t["max_rotation_separation_deg"] for t in v),
           "max_translation_separation_mm": max(t["max_translation_separation_mm"] for t in v),
           "frames_within_1e-5": sum(t["frames_within_1e-5"] for t in v), "frames_within_1e-4": sum(t["frames_within_1e-4"] for t in v),
           "frames_within_1e-3": sum(t["frames_within_1e-3"] for t in v), "frames_total": sum(t["frames"] for t in v),
           "reinits": [sum(t["reinits"][0] for t in v), sum(t["reinits"][1] for t in v)]}
    if "adds_between_tracks_mm" in v[0]:
        out.update(max_add_between_tracks_mm=max(t["add_between_tracks_mm"]["max"] for t in v),
                   max_adds_between_tracks_mm=max(t["adds_between_tracks_mm"]["max"] for t in v),
                   min_add_auc_vs_oracle_track=min(t["add_auc_vs_oracle_track"] for t in v),
                   min_adds_auc_vs_oracle_track=min(t["adds_auc_vs_oracle_track"] for t in v))
    return out


def model_points_of(mesh, se3=None):
    """the points the reference evaluates ADD / ADD-S on: the 5 mm voxel-downsampled model cloud (predict.py:131-134) when the
    product's utils are at hand, the raw vertices otherwise"""
    v = np.asarray(mesh["vertices"], np.float64)
    if se3 is not None and hasattr(se3, "utils"):
        return np.asarray(se3.utils.voxel_down_sample(v, 0.005), np.float64)
    return v


def _pool(workers):
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    return ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"))


def default_workers(jobs):
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
    try:   # cgroup quota of the box (the GPU boxes give 16 cores of a 2 x 64-core host)
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    w = max(1, min(jobs, cores // 2))
    return w, max(1, min(4, cores // w))


def camera_matrix():
    c = Fx.DATASET_INFO["camera"]
    return np.array([[c["focalX"], 0, c["centerX"]], [0, c["focalY"], c["centerY"]], [0, 0, 1.0]])


class PairRunner:
    """One worker pool + scratch directory shared by every problem set of a report: the HIP tracks run in this process as the
    problems are submitted, the oracle / control tracks in the pool; `collect` waits for them."""

    def __init__(self, se3, workers, threads):
        self.se3, self.threads = se3, threads
        self.tmp = tempfile.mkdtemp(prefix="se3tn_free_")
        self.pool = _pool(workers)
        self.n = 0

    def submit(self, problems, frames, control_keys):
        hip, futures = {}, {}
        for key, pb in problems.items():
            trk = make_hip_tracker(self.se3, pb)
            path = os.path.join(self.tmp, "t%d" % self.n)
            self.n += 1
            hip[key] = hip_track(trk, pb, frames, path)
            job = dict(problem=pb.spec(), frames=frames, numpy_rule=trk.engine.get_offset_rule(), threads=self.threads, images=path)
            futures[(key, "oracle")] = self.pool.submit(oracle_track, job)
            if key in control_keys:
                futures[(key, "control")] = self.pool.submit(oracle_track, dict(job, variant="channels_last"))
            del trk
        return hip, futures

    @staticmethod
    def collect(handle):
        hip, futures = handle
        res = {k: fut.result() for k, fut in futures.items()}
        return {key: (hip[key], res[(key, "oracle")], res.get((key, "control"))) for key in hip}

    def close(self):
        self.pool.shutdown(wait=True, cancel_futures=True)
        for fn in os.listdir(self.tmp):
            os.unlink(os.path.join(self.tmp, fn))
        os.rmdir(self.tmp)


def _blocks(tracks, pts, metrics, label=lambda key: "seed_%d" % key):
    """per-pair comparison blocks: HIP vs oracle, oracle vs its channels-last self (control), HIP vs that control track"""
    per, ctl, hc = {}, {}, {}
    for key, (h, o, c) in tracks.items():
        r = compare_tracks(h, o, pts, metrics)
        r["hip_ms_per_frame"] = round(h["ms_per_frame"], 4)
        r["oracle_seconds"] = round(o["seconds"], 1)
        per[label(key)] = r
        if c is not None:
            c = dict(c)
            hc[label(key)] = compare_tracks(h, c, pts, metrics)
            c.pop("px")                                    # (its pixel counts are against the HIP images)
            ctl[label(key)] = compare_tracks(o, c, pts, metrics)
    blk = dict(summarise(per), tracks_detail=per)
    if ctl:
        blk["control_oracle_vs_oracle_channels_last"] = dict(summarise(ctl), tracks_detail=ctl)
        blk["hip_vs_oracle_channels_last"] = dict(summarise(hc), tracks_detail=hc)
    return blk


def run_free(se3, frames=1000, seeds=(0, 1, 2), regimes=None, subdiv=5, workers=None, metrics=None, control_seeds=(0,), runner=None,
             defer=False):
    """The `free_running` block on the RANDOM-INIT stand-in: for every regime and seed, the HIP tracker's own closed loop and the
    oracle's own closed loop over the same `frames` camera frames from the same start; for `control_seeds` also the CONTROL: the
    oracle's closed loop with channels-last network inputs (same reference arithmetic, another summation order) -- how far the
    reference path separates from ITSELF.  See the module docstring."""
    from . import raster_oracle as R
    regimes = list(regimes or CL.REGIMES)
    mesh = R.icosphere(subdiv, 0.06, 0)
    metrics = metrics or getattr(se3, "metrics", None)
    pts = model_points_of(mesh, se3)
    K = camera_matrix()
    njobs = len(seeds) * len(regimes) + len(control_seeds) * len(regimes)
    nw, threads = (workers, 2) if workers else default_workers(njobs)
    t_start = time.time()
    problems = {}
    for seed in seeds:
        sd = calibrated_weights(seed, mesh, CL._frames(), K)
        for regime in regimes:
            problems[(regime, seed)] = RandomInitProblem(seed, regime, sd=sd, mesh=mesh)
    own = runner is None
    assert not (defer and own), "defer needs the caller's runner"
    runner = runner or PairRunner(se3, nw, threads)
    try:
        handle = runner.submit(problems, frames, {(r, s) for r in regimes for s in control_seeds})
        if defer:                                            # the caller collects later (other problem sets share the pool)
            return lambda: _finish_free(runner.collect(handle), frames, seeds, control_seeds, regimes, pts, metrics, runner, t_start)
        tracks = runner.collect(handle)
    finally:
        if own:
            runner.close()
    return _finish_free(tracks, frames, seeds, control_seeds, regimes, pts, metrics, runner, t_start)


def _finish_free(tracks, frames, seeds, control_seeds, regimes, pts, metrics, runner, t_start):
    nw, threads = runner.pool._max_workers, runner.threads
    out = {"what": "two INDEPENDENT closed loops per (regime, seed) over the same frames from the same start: the HIP tracker feeds back "
                   "its own pose and renders its own image A, the CPU oracle feeds back ITS own pose and renders its own image A "
                   "(oracle/free_run.py).  `control`: the oracle against ITSELF with channels-last network inputs (the same torch-CPU "
                   "arithmetic in another summation order).  Random-init weights have no restoring force (a trained tracker "
                   "re-estimates the pose from the observed frame every step): these figures are the accumulated open-loop sensitivity of "
                   "pose -> image A -> network -> pose to rounding differences, not the behaviour of a trained tracker",
           "frames": frames, "seeds": list(seeds), "control_seeds": list(control_seeds), "oracle_workers": nw,
           "oracle_threads_per_worker": threads, "regimes": {}}
    for regime in regimes:
        out["regimes"][regime] = _blocks({s: tracks[(regime, s)] for s in seeds}, pts, metrics)
    out["seconds"] = round(time.time() - t_start, 1)
    return out


def against_ground_truth(track, pb, pts, metrics):
    """ADD / ADD-S of a track against the sequence's ground truth and their AUC (eval_ycb.py:45-119: VOCap up to 0.1 m, x100)"""
    F = len(track["outs"])
    add = np.array([metrics.add(track["poses"][f], pb.ground_truth(f), pts) for f in range(1, F + 1)])
    adi = np.array([metrics.adi(track["poses"][f], pb.ground_truth(f), pts, workers=1) for f in range(1, F + 1)])
    return {"add_auc": metrics.auc(add), "adds_auc": metrics.auc(adi), "add_mm_median": float(np.median(add) * 1e3), "add_mm_max": float(add.max() * 1e3),
            "adds_mm_median": float(np.median(adi) * 1e3), "adds_mm_max": float(adi.max() * 1e3), "reinits": int(track["reinits"])}


def run_tracked(se3, frames=1000, seeds=(0, 1, 2), workers=None, metrics=None, control_seeds=(0,), weights=None, runner=None,
                defer=False, regime=None):
    """The `free_running` block on the synthetic tracking problem WITH ground truth and trained stand-in weights
    (oracle/synth_track.py, tests/golden/synth_tracker.npz): the loop of predict.py:416-420 unmodified on both sides -- started at
    the ground-truth pose of frame 0, prev_pose <- on_track(prev_pose, frame) -- plus what configs[2] reports: ADD / ADD-S AUC of each
    track against the ground truth and the Hz of the HIP track."""
    from . import synth_track as ST
    metrics = metrics or getattr(se3, "metrics", None)
    K = camera_matrix()
    njobs = len(seeds) + len(control_seeds)
    nw, threads = (workers, 2) if workers else default_workers(njobs)
    t_start = time.time()
    tmp = tempfile.mkdtemp(prefix="se3tn_seq_")
    problems = {}
    own = runner is None
    runner = runner or PairRunner(se3, nw, threads)
    nw, threads = runner.pool._max_workers, runner.threads
    try:
        for seed in seeds:                                  # the object patches of every frame (CPU renders), in the pool
            seq = ST.make_sequence(seed, frames + 1, K, runner.pool, regime=regime)
            path = os.path.join(tmp, "seq_%d.npz" % seed)
            seq.save(path)
            problems[seed] = SynthTrackProblem(seed, path, weights, regime)
        t_seq = time.time() - t_start
        handle = runner.submit(problems, frames, set(control_seeds))
    except BaseException:
        if own:
            runner.close()
        _rmtree(tmp)
        raise

    def finish():
        try:
            tracks = runner.collect(handle)
        finally:
            if own:
                runner.close()
            _rmtree(tmp)
        return _finish_tracked(tracks, problems, frames, seeds, control_seeds, se3, metrics, nw, threads, t_seq, t_start)
    return finish if defer else finish()


def _rmtree(tmp):
    if os.path.isdir(tmp):
        for fn in os.listdir(tmp):
            os.unlink(os.path.join(tmp, fn))
        os.rmdir(tmp)


def _finish_tracked(tracks, problems, frames, seeds, control_seeds, se3, metrics, nw, threads, t_seq, t_start):
    pb0 = problems[seeds[0]]
    pts = model_points_of(pb0.mesh, se3)
    out = {"what": "synthetic tracking problem with ground truth (oracle/synth_track.py): an ellipsoid with a smooth colour pattern moves "
                   "4-7 mm and %s degrees per frame in front of structured backgrounds; stand-in weights TRAINED on that problem "
                   "(%s: the pretrained YCB weights are not available offline); the loop of predict.py:"
                   "416-420 unmodified: start at the ground-truth pose of frame 0, prev_pose <- on_track(prev_pose, frame).  Two "
                   "INDEPENDENT runs of that loop -- HIP tracker / CPU oracle -- and the oracle against itself (channels-last control)"
                   % ("3-4.5" if pb0.regime == "ycbineoat_30deg" else "1-1.5", os.path.join("tests", "golden", os.path.basename(pb0.weights_path))),
           "regime": pb0.regime, "frames": frames, "seeds": list(seeds), "control_seeds": list(control_seeds), "oracle_workers": nw,
           "oracle_threads_per_worker": threads, "trans_normalizer": pb0.tn, "rot_normalizer_deg": round(pb0.rn * 180 / np.pi, 3),
           "weights": pb0.weights_info, "sequence_seconds": round(t_seq, 1)}
    out.update(_blocks(tracks, pts, metrics))
    gt = {}
    for seed, (h, o, c) in tracks.items():
        g = {"hip": against_ground_truth(h, problems[seed], pts, metrics), "oracle": against_ground_truth(o, problems[seed], pts, metrics)}
        if c is not None:
            g["oracle_channels_last"] = against_ground_truth(c, problems[seed], pts, metrics)
        g["adds_auc_hip_minus_oracle"] = g["hip"]["adds_auc"] - g["oracle"]["adds_auc"]
        g["add_auc_hip_minus_oracle"] = g["hip"]["add_auc"] - g["oracle"]["add_auc"]
        gt["seed_%d" % seed] = g
    out["against_ground_truth"] = gt
    out["hz_hip"] = round(1000.0 / float(np.median([h["ms_per_frame"] for h, _, _ in tracks.values()])), 1)
    out["seconds"] = round(time.time() - t_start, 1)
    return out


TRACKED_REGIMES = (("ycbineoat_30deg", "synthetic_tracking_trained_weights"), ("ycb_video_5deg", "synthetic_tracking_trained_weights_5deg"))


def run_report(se3, frames_tracked=1000, frames_random=1000, seeds=(0, 1, 2), control_seeds=(0,), workers=None):
    """Every problem set through ONE worker pool (bench.py's `track.free_running`): the synthetic tracking problem first, under each
    normaliser regime that has a trained fixture (its sequences are rendered in the still-idle pool), then the random-init stand-in; all
    sets of oracle tracks run side by side."""
    have = [(r, k) for r, k in TRACKED_REGIMES if os.path.exists(default_synth_weights(r))] if frames_tracked > 0 else []
    njobs = (len(seeds) + len(control_seeds)) * (2 * (frames_random > 0) + len(have))
    nw, threads = (workers, 2) if workers else default_workers(max(njobs, 1))
    runner = PairRunner(se3, nw, threads)
    out, fins, fin_r = {}, [], None
    try:
        for regime, key in have:
            fins.append((key, run_tracked(se3, frames_tracked, seeds, control_seeds=control_seeds, runner=runner, defer=True, regime=regime)))
        if frames_tracked > 0 and not have:
            out["synthetic_tracking_trained_weights"] = {"skipped": "tests/golden/synth_tracker.npz not found"}
        if frames_random > 0:
            fin_r = run_free(se3, frames_random, seeds, control_seeds=control_seeds, runner=runner, defer=True)
        for key, fin in fins:
            out[key] = fin()
        if fin_r is not None:
            out["random_init"] = fin_r()
    finally:
        runner.close()
    return out
