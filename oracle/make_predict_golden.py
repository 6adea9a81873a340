"""BASELINE configs[0] for real: the reference's OWN `predict.Tracker` (predict.py:127-296) -- constructor, render_window, on_track --
executed end to end in the build container, UNMODIFIED: torch on the CPU for the network (configs[0]: "PyTorch CPU forward via
predict.py"), the reference's VispyRenderer on a real OpenGL implementation (SwiftShader, oracle/swiftshader_gl.py), the reference's
Utils / datasets / data_augmentation for everything else.  Output: tests/golden/predict_tracker.npz (TEST INFRASTRUCTURE ONLY).

    python -m oracle.make_predict_golden [out_dir]

What had to be supplied around the unmodified files (nothing on the arithmetic path is this repo's):
  * stand-in modules for imports that are not installable offline and are not used by Tracker: pyrender, transformations,
    torchvision (empty); `trimesh.load` (-> an object with .vertices, read by this repo's PLY reader; only the point cloud /
    object_width use it and dataset_info carries object_width); `open3d` point cloud with voxel_down_sample (Tracker.object_cloud,
    unused by on_track); the vispy / PyOpenGL / plyfile stand-ins over SwiftShader;
  * `cv2`: resize(INTER_NEAREST) and Rodrigues as restated in oracle/se3_oracle.py (OpenCV is absent: those two rules stay
    "unpinned"), and no-op imshow / waitKey (the GUI of predict.py:286-290);
  * `.cuda()` -> identity (this container has no GPU; the reference moves the model and the two input tensors to the GPU,
    predict.py:157,267-268);
  * a checkpoint file written with torch.save({'state_dict': ...}) and mean / std arrays, as predict.py:151-156,657-658 load them."""
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

from . import fixtures as Fx
from . import ref_shims
from . import se3_oracle as O
from . import swiftshader_gl as SG
from .make_gl_golden import write_ply

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEAD_GAIN = 0.002
FRAMES = 6
OBJECT_WIDTH = 150.0
MESH = (3, 0.06, 5)     # icosphere(subdiv, radius, seed)


def load_predict():
    ref_shims.install()
    SG.install_stubs()
    # other tests of the same session may have parked empty stand-ins for matplotlib (eval_ycb imports pyplot without using it):
    # predict.py imports mpl_toolkits.mplot3d, which needs the real package
    for name in [n for n in sys.modules if n == "matplotlib" or n.startswith("matplotlib.") or n.startswith("mpl_toolkits")]:
        if not getattr(sys.modules[name], "__file__", None):
            del sys.modules[name]
    from . import ply_io as U     # numpy stand-in loaders, independent of the product package

    # open3d: the point cloud Tracker.__init__ builds (predict.py:131-133)
    o3d = types.ModuleType("open3d")

    class PointCloud:
        def __init__(self):
            self.points = np.zeros((0, 3)); self.colors = np.zeros((0, 3))

        def voxel_down_sample(self, voxel_size):
            out = PointCloud()
            out.points = U.voxel_down_sample(np.asarray(self.points), voxel_size)
            return out

        def transform(self, T):
            T = np.asarray(T, np.float64)
            self.points = np.asarray(self.points) @ T[:3, :3].T + T[:3, 3]
            return self
    o3d.geometry = types.SimpleNamespace(PointCloud=PointCloud)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.array(a, np.float64))
    sys.modules["open3d"] = o3d
    tm = types.ModuleType("trimesh")
    tm.load = lambda path: types.SimpleNamespace(vertices=U.ply_vertices(path))
    sys.modules["trimesh"] = tm
    sys.modules["pyrender"] = types.ModuleType("pyrender")
    cv2 = sys.modules["cv2"]
    cv2.imshow = lambda *a, **k: None
    cv2.waitKey = lambda *a, **k: -1
    cv2.cvtColor = lambda img, code: img[..., ::-1]
    cv2.COLOR_RGB2BGR = cv2.COLOR_BGR2RGB = 4
    torch.nn.Module.cuda = lambda self, device=None: self
    torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("Utils", "eval_ycb", "offscreen_renderer", "vispy_renderer", "predict", "datasets", "data_augmentation", "se3_tracknet"):
        sys.modules.pop(name, None)
    return importlib.import_module("predict")


def run(tmp):
    torch.set_num_threads(max(1, os.cpu_count() or 1))     # as oracle/make_golden.py: one thread count -> one summation order
    predict = load_predict()
    mean, std = Fx.mean_std(0)
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    ckpt = os.path.join(tmp, "model_best_val.pth.tar")
    torch.save({"state_dict": sd, "epoch": 1}, ckpt)
    mesh = Fx.icosphere(*MESH)
    ply = os.path.join(tmp, "model.ply")
    write_ply(ply, mesh)
    info = dict(Fx.DATASET_INFO, object_width=OBJECT_WIDTH)
    trk = predict.Tracker(info, mean, std, ckpt, model_path=ply)           # the reference's class, unmodified
    assert type(trk.renderer).__name__ == "VispyRenderer" and type(trk.model).__name__ == "Se3TrackNet"
    P = Fx.pose(3, (0.02, -0.01, 0.8))
    out = {"pose0": P.copy(), "K": trk.K.copy(), "object_width": np.float64(trk.object_width)}
    poses, rgbAs, depthAs = [], [], []
    for f in range(FRAMES):
        rgb, depth = Fx.structured_frame(700 + f)
        rgbA, depthA = trk.render_window(P)                                 # what on_track renders for this pose (predict.py:246)
        rgbAs.append(np.array(rgbA)); depthAs.append(np.array(depthA))
        P = trk.on_track(P, rgb, depth, gt_A_in_cam=np.eye(4), gt_B_in_cam=np.eye(4), debug=False, samples=1)
        poses.append(np.array(P))
    out.update(poses=np.array(poses), rgbA=np.array(rgbAs), depthA=np.array(depthAs), frame_cnt=np.int64(trk.frame_cnt))
    return out


def main(out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden")
    with tempfile.TemporaryDirectory() as tmp:
        g = run(tmp)
    np.savez_compressed(os.path.join(out_dir, "predict_tracker.npz"), **g)
    print("predict_tracker.npz: %d frames of predict.Tracker.on_track; last pose\n%s" % (len(g["poses"]), g["poses"][-1]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
