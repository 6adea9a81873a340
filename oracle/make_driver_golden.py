"""The reference's own sequence DRIVER, end to end: `predict.predictSequenceYcbInEOAT()` (predict.py:578-626), imported unmodified and
run in the build container on a synthetic YCBInEOAT video (rgb/ depth_filled/ annotated_poses/), with the reference's Tracker
(30-degree rotation normaliser, :586), its VispyRenderer on a real GL implementation and torch-CPU -- see oracle/make_predict_golden.py
for the stand-ins.  Stores the result files' contents (one 4x4 per %07d.txt) and, recorded by a pass-through wrapper around
Tracker.render_window, the image A of every frame: tests/golden/driver_ycbineoat.npz (TEST INFRASTRUCTURE ONLY).

    python -m oracle.make_driver_golden [out_dir]

Additional stand-ins the driver needs (GUI / file IO only): cv2.imread (PIL; IMREAD_UNCHANGED keeps uint16), cv2.circle / putText /
imshow / waitKey (no-ops), cv2.resize of the visualisation frame (unused result)."""
import glob
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

from . import fixtures as Fx
from . import se3_oracle as O
from .make_gl_golden import write_ply
from .make_predict_golden import HEAD_GAIN, MESH, OBJECT_WIDTH, load_predict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIDEO = "mustard0"
N_FRAMES = 7
FRAME_HW = (240, 320)


def make_video(root):
    """<root>/<VIDEO>/{rgb,depth_filled,annotated_poses}: structured frames (smaller than the 480 x 640 the intrinsics describe: part of
    every crop window is zero padding, as at a frame border)."""
    d = os.path.join(root, VIDEO)
    for sub in ("rgb", "depth_filled", "annotated_poses"):
        os.makedirs(os.path.join(d, sub), exist_ok=True)
    for i in range(N_FRAMES):
        rgb, depth = Fx.structured_frame(900 + i, *FRAME_HW)
        Image.fromarray(rgb).save(os.path.join(d, "rgb", "%07d.png" % i))
        Image.fromarray(depth).save(os.path.join(d, "depth_filled", "%07d.png" % i))
        np.savetxt(os.path.join(d, "annotated_poses", "%07d.txt" % i), Fx.pose(3, (0.01 + 0.002 * i, -0.03, 0.55)))
    return d


def run(tmp):
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    predict = load_predict()
    cv2 = sys.modules["cv2"]

    def imread(path, flags=1):
        a = np.array(Image.open(path))
        return a if flags == cv2.IMREAD_UNCHANGED or a.ndim == 2 else a[..., ::-1].copy()
    cv2.imread = imread
    cv2.circle = cv2.putText = lambda *a, **k: None
    cv2.FONT_HERSHEY_SIMPLEX = 0
    nearest = cv2.resize
    cv2.resize = lambda img, dsize, interpolation=None, **k: (nearest(img, dsize, interpolation=cv2.INTER_NEAREST)
                                                              if interpolation == cv2.INTER_NEAREST else np.zeros(dsize[::-1] + (3,), np.uint8))
    mean, std = Fx.mean_std(0)
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    ckpt = os.path.join(tmp, "model_best_val.pth.tar")
    torch.save({"state_dict": sd}, ckpt)
    ply = os.path.join(tmp, "model.ply")
    write_ply(ply, Fx.icosphere(*MESH))
    video = make_video(tmp)
    outdir = os.path.join(tmp, "res", VIDEO) + "/"
    # the globals predict.py's __main__ block sets (predict.py:628-660)
    predict.args = types.SimpleNamespace(YCBInEOAT_dir=video)
    predict.dataset_info = dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH)
    predict.images_mean, predict.images_std = mean, std
    predict.ckpt_dir, predict.model_path, predict.outdir = ckpt, ply, outdir
    rec = {"rgbA": [], "depthA": [], "poses_in": []}
    orig = predict.Tracker.render_window

    def recording(self, ob2cam):
        rgb, depth = orig(self, ob2cam)
        rec["rgbA"].append(np.array(rgb)); rec["depthA"].append(np.array(depth)); rec["poses_in"].append(np.array(ob2cam))
        return rgb, depth
    predict.Tracker.render_window = recording
    try:
        predict.predictSequenceYcbInEOAT()
    finally:
        predict.Tracker.render_window = orig
        cv2.resize = nearest
    files = sorted(glob.glob(outdir + "*.txt"))
    poses = np.array([np.loadtxt(f) for f in files])
    # on_track renders twice per frame (image A for prev_pose, then the visualisation of the estimate, predict.py:246,284): keep the first
    assert len(rec["rgbA"]) == 2 * len(files)
    return {"files": np.array([os.path.basename(f) for f in files]), "poses": poses, "rgbA": np.array(rec["rgbA"][0::2]),
            "depthA": np.array(rec["depthA"][0::2]), "poses_in": np.array(rec["poses_in"][0::2])}


def main(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as tmp:
        g = run(tmp)
    np.savez_compressed(os.path.join(out_dir, "driver_ycbineoat.npz"), **g)
    print("driver_ycbineoat.npz: %d result files %s .. %s" % (len(g["files"]), g["files"][0], g["files"][-1]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
