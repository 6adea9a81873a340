"""Generate tests/golden/*.npz by running the REFERENCE's own code (imported unmodified
from /root/reference through oracle/ref_shims.py) on seeded inputs.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (the reference tree is not
on the GPU box); the produced fixtures are committed.  Inputs and weights are *not*
stored (54 MB): they are regenerated from seeds by oracle/fixtures.py and
oracle/se3_oracle.make_state_dict; each golden carries fingerprints of them.
"""
import os
import sys

import numpy as np
import torch

from . import fixtures as Fx
from . import ref_shims
from . import se3_oracle as O

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
SUB = 7  # stride of the stored sub-sample of big tensors
ON_TRACK_HEAD_GAIN = 0.002


def ref_model(ref, sd):
    model = ref.se3_tracknet.Se3TrackNet(image_size=176)
    missing = model.load_state_dict(sd, strict=True)  # pins the key/shape surface
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.eval()


def gold_network(ref):
    """Se3TrackNet.forward (se3_tracknet.py:81-112) on N(0,1) inputs, batch 3, plus the
    pre-tanh logits and sub-sampled intermediate feature maps (captured with hooks)."""
    sd = O.make_state_dict(0)
    model = ref_model(ref, sd)
    A, B = Fx.net_inputs(1, 3)
    caps = {}
    hooks = []
    for name in ("convA1", "poolA1", "convA2", "convB1", "convB2", "convB3", "convAB1",
                 "convAB2", "trans_conv1", "trans_conv2", "rot_conv1", "rot_conv2"):
        hooks.append(getattr(model, name).register_forward_hook(
            lambda m, i, o, name=name: caps.__setitem__(name, o.detach().clone())))
    hooks.append(model.trans_out[0].register_forward_hook(
        lambda m, i, o: caps.__setitem__("trans_logit", o.detach().clone())))
    hooks.append(model.rot_out[0].register_forward_hook(
        lambda m, i, o: caps.__setitem__("rot_logit", o.detach().clone())))
    with torch.no_grad():
        out = model(A, B)
    for h in hooks:
        h.remove()
    d = dict(trans=out["trans"].numpy(), rot=out["rot"].numpy(),
             trans_logit=caps["trans_logit"].numpy(), rot_logit=caps["rot_logit"].numpy(),
             A_fp=np.array([float(A.double().sum()), float(A[0, 0, 0, 0])]),
             sd_fp=np.array([float(sum(v.double().sum() for v in sd.values()))]),
             sub=np.array(SUB))
    for k in ("convA1", "poolA1", "convA2", "convB1", "convB2", "convB3", "convAB1",
              "convAB2", "trans_conv1", "trans_conv2", "rot_conv1", "rot_conv2"):
        d["act_" + k] = caps[k].numpy()[:, ::SUB, ::SUB, ::SUB].copy()
    d["feature"] = out["feature"].numpy()[:, ::SUB, ::SUB, ::SUB].copy()
    np.savez_compressed(os.path.join(OUT, "network_n3.npz"), **d)
    print("network_n3: trans", d["trans"], "rot", d["rot"])

    # second weight seed + large-magnitude inputs (normalised real inputs are O(10^2),
    # SURVEY.md section 5): only logits/outputs
    sd2 = O.make_state_dict(7, head_gain=0.002)
    model2 = ref_model(ref, sd2)
    A2, B2 = Fx.net_inputs(11, 2, scale=40.0)
    caps2 = {}
    h1 = model2.trans_out[0].register_forward_hook(lambda m, i, o: caps2.__setitem__("t", o.detach().clone()))
    h2 = model2.rot_out[0].register_forward_hook(lambda m, i, o: caps2.__setitem__("r", o.detach().clone()))
    with torch.no_grad():
        out2 = model2(A2, B2)
    h1.remove(); h2.remove()
    np.savez_compressed(os.path.join(OUT, "network_big_n2.npz"), trans=out2["trans"].numpy(),
                        rot=out2["rot"].numpy(), trans_logit=caps2["t"].numpy(),
                        rot_logit=caps2["r"].numpy())
    print("network_big_n2: logits", caps2["t"].numpy(), caps2["r"].numpy())


# (file, weight seed, head gain, input seed, n, input scale): logits / outputs only.  n >= 8 so that the
# engine's DEFAULT large-batch algorithm (Winograd F(4x4,3x3) from SE3TN_WINOGRAD_DEFAULT_MIN_BATCH = 6
# pairs) is what the GPU tests compare with these reference-made numbers; x40 / x150 inputs are the
# magnitude real std.npy files produce; n = 64 is BASELINE configs[1]'s batch.
LARGE_CASES = [
    ("network_big_n8", 7, 0.002, 12, 8, 40.0),
    ("network_huge_n8", 9, 0.0005, 13, 8, 150.0),
    ("network_n64", 0, 0.05, 5, 64, 1.0),
    # n = 16 >= SE3TN_WINOGRAD_TILE6_MIN_BATCH (14): SE3TN_WINOGRAD_TILE_AUTO runs F(6x6,3x3) here -- the engine's default algorithm for
    # BASELINE's batch -- on the same large-magnitude inputs
    ("network_big_n16", 7, 0.002, 14, 16, 40.0),
    ("network_huge_n16", 9, 0.0005, 15, 16, 150.0),
]


def gold_network_large(ref):
    for fname, wseed, gain, iseed, n, scale in LARGE_CASES:
        sd = O.make_state_dict(wseed, head_gain=gain)
        model = ref_model(ref, sd)
        A, B = Fx.net_inputs(iseed, n, scale=scale)
        caps = {}
        h1 = model.trans_out[0].register_forward_hook(lambda m, i, o: caps.__setitem__("t", o.detach().clone()))
        h2 = model.rot_out[0].register_forward_hook(lambda m, i, o: caps.__setitem__("r", o.detach().clone()))
        with torch.no_grad():
            out = model(A, B)
        h1.remove(); h2.remove()
        np.savez_compressed(os.path.join(OUT, fname + ".npz"), trans=out["trans"].numpy(), rot=out["rot"].numpy(),
                            trans_logit=caps["t"].numpy(), rot_logit=caps["r"].numpy(),
                            A_fp=np.array([float(A.double().sum()), float(A[0, 0, 0, 0])]),
                            case=np.array([wseed, gain, iseed, n, scale]))
        print(fname, "max |logit| %.3f  max |tanh| %.3f" % (
            max(float(caps["t"].abs().max()), float(caps["r"].abs().max())),
            max(float(out["trans"].abs().max()), float(out["rot"].abs().max()))))


PRE_CASES = [
    # (name, frame_seed, translation, object_width)
    ("center", 3, (0.05, -0.02, 0.8), 250.0),       # bbox inside the frame
    ("topleft", 4, (-0.22, -0.17, 0.75), 230.0),    # crosses the top/left border
    ("botright", 5, (0.21, 0.16, 0.7), 260.0),      # crosses the bottom/right border
    ("near", 6, (0.0, 0.0, 0.32), 250.0),           # crop larger than the frame height
    ("far", 8, (0.01, 0.03, 1.9), 120.0),           # crop smaller than 176 (up-sampling)
]


def gold_preprocess(ref):
    """compute_bbox (Utils.py:302) -> crop_bbox (Utils.py:320) -> TrackDataset.processData
    with OffsetDepth/NormalizeChannels/ToTensor (predict.py:189, datasets.py:115)."""
    U, DA, DS = ref.Utils, ref.data_augmentation, ref.datasets
    mean, std = Fx.mean_std(0)
    post = U.Compose([DA.OffsetDepth(), DA.NormalizeChannels(mean, std), DA.ToTensor()])
    ds = DS.TrackDataset('', 'eval', mean, std, None, None, post, Fx.DATASET_INFO)
    d = {}
    for name, fseed, t, width in PRE_CASES:
        rgb, depth = Fx.synthetic_frame(fseed)
        P = Fx.pose(fseed, t)
        rgbA, depthA = Fx.synthetic_render(fseed + 100, t[2])
        bb = U.compute_bbox(P, Fx.K_YCB, width, scale=(1000, 1000, 1000))
        rgbB, depthB = U.crop_bbox(rgb, depth, bb, (176, 176))
        sample = ds.processData(rgbA, depthA, P, rgbB, depthB, np.eye(4))[0]
        a, b = sample[0].numpy(), sample[1].numpy()
        assert a.dtype == np.float32 and a.shape == (4, 176, 176)
        d[name + "_bbox"] = bb
        d[name + "_rgbB_sha"] = np.array(Fx.sha(rgbB))
        d[name + "_depthB_sha"] = np.array(Fx.sha(depthB))
        d[name + "_dataA_sha"] = np.array(Fx.sha(a))
        d[name + "_dataB_sha"] = np.array(Fx.sha(b))
        d[name + "_dataA_sub"] = a[:, ::SUB, ::SUB].copy()
        d[name + "_dataB_sub"] = b[:, ::SUB, ::SUB].copy()
        d[name + "_frame_sha"] = np.array(Fx.sha(rgb) + Fx.sha(depth))
        print("preprocess", name, "bbox", bb.tolist())
    np.savez_compressed(os.path.join(OUT, "preprocess.npz"), **d)


def gold_pose_update(ref):
    """TrackDataset.processPredict (datasets.py:159-175) incl. zero rotation and both
    normaliser sets (predict.py:128 YCB-Video, predict.py:586 YCBInEOAT)."""
    DS = ref.datasets
    rng = np.random.default_rng(21)
    poses, trans, rots, outs, norms = [], [], [], [], []
    for i in range(12):
        P = Fx.pose(50 + i, tuple(rng.uniform(-0.3, 0.3, 2)) + (float(rng.uniform(0.4, 1.2)),))
        t = rng.uniform(-1, 1, 3).astype(np.float32)
        r = rng.uniform(-1, 1, 3).astype(np.float32)
        if i == 0:
            r[:] = 0
        if i == 1:
            r[:] = (1e-20, 0, 0)
        tn, rn = (0.03, 5 * np.pi / 180) if i % 2 == 0 else (0.03, 30 * np.pi / 180)
        ds = DS.TrackDataset('', 'eval', None, None, None, None, None, Fx.DATASET_INFO,
                             trans_normalizer=tn, rot_normalizer=rn)
        outs.append(ds.processPredict(P, (t, r)))
        poses.append(P); trans.append(t); rots.append(r); norms.append((tn, rn))
    np.savez_compressed(os.path.join(OUT, "pose_update.npz"), A=np.array(poses), trans=np.array(trans),
                        rot=np.array(rots), norm=np.array(norms), B=np.array(outs))
    print("pose_update: 12 cases")


def gold_on_track(ref):
    """The arithmetic of Tracker.on_track (predict.py:217-296) composed from the reference's
    inner functions (predict.py itself cannot be imported offline: open3d/vispy/pyrender),
    iterated over 3 frames with pose feedback; rendered A is synthetic."""
    U, DA, DS = ref.Utils, ref.data_augmentation, ref.datasets
    sd = O.make_state_dict(0, head_gain=ON_TRACK_HEAD_GAIN)  # keeps tanh out of saturation
    model = ref_model(ref, sd)
    mean, std = Fx.mean_std(0)
    post = U.Compose([DA.OffsetDepth(), DA.NormalizeChannels(mean, std), DA.ToTensor()])
    ds = DS.TrackDataset('', 'eval', mean, std, None, None, post, Fx.DATASET_INFO,
                         trans_normalizer=0.03, rot_normalizer=5 * np.pi / 180)
    P = Fx.pose(3)
    poses, bbs, trs, rts = [P.copy()], [], [], []
    for f in range(3):
        rgb, depth = Fx.synthetic_frame(30 + f)
        rgbA, depthA = Fx.synthetic_render(130 + f, P[2, 3])
        bb = U.compute_bbox(P, Fx.K_YCB, 250.0, scale=(1000, 1000, 1000))
        rgbB, depthB = U.crop_bbox(rgb, depth, bb, (176, 176))
        sample = ds.processData(rgbA, depthA, P, rgbB, depthB, np.eye(4))[0]
        dataA = sample[0].unsqueeze(0).float(); dataB = sample[1].unsqueeze(0).float()
        with torch.no_grad():
            pred = model(dataA, dataB)
        t = pred["trans"][0].data.cpu().numpy(); r = pred["rot"][0].data.cpu().numpy()
        P = ds.processPredict(P, (t, r))
        poses.append(P.copy()); bbs.append(bb); trs.append(t); rts.append(r)
    np.savez_compressed(os.path.join(OUT, "on_track.npz"), poses=np.array(poses), bbox=np.array(bbs),
                        trans=np.array(trs), rot=np.array(rts))
    print("on_track: poses[-1]\n", poses[-1])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ref = ref_shims.load()
    jobs = {"network": gold_network, "network_large": gold_network_large, "preprocess": gold_preprocess,
            "pose_update": gold_pose_update, "on_track": gold_on_track}
    # `python -m oracle.make_golden network_large` regenerates one family only
    for name in (sys.argv[1:] or list(jobs)):
        jobs[name](ref)
    with open(os.path.join(OUT, "PROVENANCE.txt"), "w") as f:
        f.write("generated by `python -m oracle.make_golden` from the reference tree at %s\n"
                "torch %s numpy %s; cv2.resize(NEAREST)/cv2.Rodrigues via oracle/ref_shims.py shim "
                "(OpenCV absent offline -> parity unpinned for those two rules)\n"
                % (ref_shims.REFERENCE_ROOT, torch.__version__, np.__version__))


if __name__ == "__main__":
    sys.exit(main())
