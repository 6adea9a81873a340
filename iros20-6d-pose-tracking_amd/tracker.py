"""Drop-in for the reference's ``predict.Tracker`` (predict.py:127-296): same constructor
arguments, ``on_track`` signature / return value and the attributes callers read (``K``,
``object_cloud``, ``object_width``, ``dataset`` with processData / processPredict, callable ``model``).  The arithmetic of on_track runs on the GPU
through the C ABI: se3tn_render (image A, predict.py:193-215) -> se3tn_preprocess -> se3tn_infer (network + pose update); only
compute_bbox is host float64 exactly as the reference.  Given a model file the constructor builds the HIP rasteriser itself
(``HipRenderer``: byte-identical to the reference's VispyRenderer on the GL implementation the goldens were rendered on); a
renderer object with the reference's protocol can be injected instead (``renderer.render(ob_in_cam, K, window) -> rgb u8,
depth u16`` or the full-frame ``render([ob_in_cam])`` of offscreen_renderer.Renderer)."""
import numpy as np
import torch

from . import utils as U
from .dataset import TrackDataset
from .engine import Engine, NHWC
from .se3_tracknet import Se3TrackNet


def _depth_u16(depth, rgb):
    """HxW depth in millimetres as the int16-viewed uint16 array the engine takes.  The reference casts with
    .astype(np.uint16) (predict.py:410-412 callers); a wider dtype must be CONVERTED, never byte-reinterpreted."""
    d = np.asarray(depth)
    if d.shape != np.asarray(rgb).shape[:2]:
        raise ValueError("depth %s does not match rgb %s" % (d.shape, np.asarray(rgb).shape))
    return np.ascontiguousarray(d, dtype=np.uint16).view(np.int16)


def _is_full_frame_renderer(renderer):
    """offscreen_renderer.Renderer protocol (predict.py:209-213): ``render([ob2cam]) -> (rgb HxWx3, depth HxW metres)``.
    Detected structurally -- the reference's class has no marker attribute: an explicit ``full_frame`` attribute wins,
    otherwise a ``render`` that takes exactly ONE positional argument (the pose list) is the full-frame protocol, one that
    takes (ob2cam, K, window) is the window protocol."""
    flag = getattr(renderer, "full_frame", None)
    if flag is not None:
        return bool(flag)
    import inspect
    try:
        params = [p for p in inspect.signature(renderer.render).parameters.values()
                  if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    except (TypeError, ValueError):
        return False
    required = [p for p in params if p.default is p.empty]
    return len(required) == 1 and len(params) < 3


class Tracker:
    def __init__(self, dataset_info, images_mean, images_std, ckpt_dir, model_path=None,
                 trans_normalizer=0.03, rot_normalizer=5 * np.pi / 180, renderer=None, device=0,
                 max_samples=8, use_graphs=False):
        self.dataset_info = dataset_info
        self.image_size = (dataset_info['resolution'], dataset_info['resolution'])
        self.object_cloud = None
        if model_path is not None:
            pts = U.load_model_points(model_path)
            self.object_cloud = U.PointCloud(U.voxel_down_sample(pts, 0.005))
        if 'object_width' not in dataset_info:
            if self.object_cloud is None:
                raise ValueError("dataset_info has no 'object_width' and no model_path was given")
            object_max_width = U.compute_obj_max_width(self.object_cloud.points)
            self.object_width = object_max_width + dataset_info['boundingbox'] / 100 * object_max_width
        else:
            self.object_width = dataset_info['object_width']
        self.mean = np.asarray(images_mean, np.float64)
        self.std = np.asarray(images_std, np.float64)
        cam = dataset_info['camera']
        self.K = np.array([cam['focalX'], 0, cam['centerX'], 0, cam['focalY'], cam['centerY'], 0, 0, 1]).reshape(3, 3)

        # checkpoint surface: torch.load(path)['state_dict'] (predict.py:151-156) or a dict
        checkpoint = torch.load(ckpt_dir, map_location="cpu") if isinstance(ckpt_dir, str) else ckpt_dir
        sd = checkpoint['state_dict'] if 'state_dict' in checkpoint else checkpoint
        self.engine = Engine(device, max_samples)
        self.engine.load_state_dict(sd)
        self.engine.set_normalization(self.mean, self.std)
        # start-up reservation for this camera: full-frame z-buffer / fill_depth scratch, Winograd planes -- no stream-ordered
        # call allocates afterwards (se3tn_reserve)
        self.engine.reserve(int(cam['height']), int(cam['width']))
        self.trans_normalizer = float(trans_normalizer)
        self.rot_normalizer = float(rot_normalizer)
        self.engine.set_normalizers(self.trans_normalizer, self.rot_normalizer)
        # predict.py:270-271 `prediction = self.model(dataA, dataB)`: float32 CUDA [N,4,176,176] -> the reference's dict
        self.model = Se3TrackNet.from_engine(self.engine)
        # predict.py:189-191: the inner pre / post boundary (processData / processPredict)
        self.dataset = TrackDataset(self.engine, self.mean, self.std, dataset_info, self.trans_normalizer,
                                    self.rot_normalizer)
        self.renderer = renderer
        if renderer is None and model_path is not None and dataset_info.get('renderer') == 'pyrenderer':
            # predict.py:161-164: textured .obj through the pyrender-style full-frame renderer
            assert '.obj' in model_path
            from .renderer import HipRenderer
            self.renderer = HipRenderer(self.engine, model_path, mode="pyrender", frame_size=(cam['height'], cam['width']))
        elif renderer is None and model_path is not None and model_path.endswith(".ply"):
            # the reference builds a VispyRenderer from the .ply here (predict.py:180-182); ours is the
            # HIP rasteriser -- only if the file has faces (the repo's bunny fixture has none)
            from .renderer import HipRenderer
            mesh = U.load_ply_mesh(model_path)
            if len(mesh["faces"]) > 0:
                self.renderer = HipRenderer(self.engine, mesh)
        self.prev_rgb = None
        self.prev_depth = None
        self.frame_cnt = 0
        self._one_call_state = None
        self._batch_state = None
        self.one_call = True     # on_track through se3tn_on_track when the built-in rasteriser renders image A (False: step by step)
        self.errs = []
        dev = "cuda:%d" % device
        self._dev = dev
        self._poseA = torch.empty((max_samples, 16), dtype=torch.float64, device=dev)
        # poseB | trans | rot live in ONE device buffer: what predict.py:275-276 reads back per frame is one D2H copy + one sync
        ms = int(max_samples)
        self._out = torch.zeros(ms * (128 + 12 + 12), dtype=torch.uint8, device=dev)
        self._poseB = self._out[:ms * 128].view(torch.float64).view(ms, 16)
        self._trans = self._out[ms * 128:ms * 140].view(torch.float32).view(ms, 3)
        self._rot = self._out[ms * 140:ms * 152].view(torch.float32).view(ms, 3)
        self.last_prediction = None
        # optional hipGraph replay of the ~20 dependent launches of a frame on a dedicated stream (the
        # null stream cannot be captured).  Off by default: measured 0.213 ms/frame with vs 0.194 without
        # (profiles/EXPERIMENTS.md item 54: the kernels are 5-15 us each and the eager launches already run ahead of the device)
        self._stream = torch.cuda.Stream(device=dev) if use_graphs else None
        if use_graphs:
            self.engine.enable_graphs(True)

    def _read_back(self, n):
        """(poseB [n,4,4] float64, trans [n,3], rot [n,3] float32) of the last engine call: one device-to-host copy."""
        ms = self.engine.max_batch
        host = self._out.cpu().numpy()
        poseB = host[:ms * 128].view(np.float64).reshape(ms, 4, 4)[:n].copy()
        trans = host[ms * 128:ms * 140].view(np.float32).reshape(ms, 3)[:n].copy()
        rot = host[ms * 140:ms * 152].view(np.float32).reshape(ms, 3)[:n].copy()
        return poseB, trans, rot

    def render_window(self, ob2cam):
        """predict.py:193-215.  Three renderer protocols, in the reference's order:
          * VispyRenderer-like objects (``update_cam_mat`` + ``render_image``): driven exactly as
            predict.py:201-208 does -- y-flipped bbox (scale (1000,-1000,1000)), ``update_cam_mat(K, left,
            right, bottom, top)``, ``render_image(ob2cam_gl)``;
          * the HIP rasteriser and any injected ``render(ob2cam, K, window)`` object: the same y-flipped
            window (left, top, right, bottom) is passed (round 1 passed the plain crop window: INTEGRATION.md);
          * full-frame renderers in the style of offscreen_renderer.Renderer (``render([ob2cam])`` ->
            rgb HxWx3, depth HxW metres; predict.py:209-213): depth -> uint16 mm, then the crop + NEAREST
            resize of crop_bbox on the device (se3tn_crop_raw)."""
        if self.renderer is None:
            raise RuntimeError("Tracker.render_window: no renderer injected (rendering is outside the HIP hot path)")
        from .renderer import HipRenderer
        ob2cam = np.asarray(ob2cam, np.float64)
        if isinstance(self.renderer, HipRenderer) and self.renderer.full_frame:     # predict.py:209-213
            rgb_d, dep_d = self.renderer.render_frame_device(ob2cam, self.K)
            bbox = U.compute_bbox(ob2cam, self.K, self.object_width, scale=(1000, 1000, 1000))
            return self.engine.crop_raw(rgb_d, dep_d, U.crop_window(bbox))
        win = HipRenderer.gl_window(ob2cam, self.K, self.object_width)      # left, top, right, bottom (GL image)
        if hasattr(self.renderer, "update_cam_mat") and hasattr(self.renderer, "render_image"):
            glcam_in_cvcam = np.diag([1.0, -1.0, -1.0, 1.0])
            self.renderer.update_cam_mat(self.K, win[0], win[2], win[3], win[1])
            return self.renderer.render_image(np.linalg.inv(glcam_in_cvcam).dot(ob2cam))
        if _is_full_frame_renderer(self.renderer):
            rgb, depth = self.renderer.render([ob2cam])
            depth = (np.asarray(depth) * 1000).astype(np.uint16)
            bbox = U.compute_bbox(ob2cam, self.K, self.object_width, scale=(1000, 1000, 1000))
            return self.engine.crop_raw(rgb, depth, U.crop_window(bbox))
        return self.renderer.render(ob2cam, self.K, win)

    def on_track(self, prev_pose, current_rgb, current_depth, gt_A_in_cam=None, gt_B_in_cam=None,
                 debug=False, samples=1):
        """predict.py:217-296.  current_rgb HxWx3 uint8 RGB, current_depth HxW uint16 mm,
        prev_pose 4x4 object-in-camera (metres).  Returns the 4x4 float64 pose estimate."""
        if self._stream is not None and torch.cuda.current_stream() != self._stream:
            with torch.cuda.stream(self._stream):
                return self.on_track(prev_pose, current_rgb, current_depth, gt_A_in_cam, gt_B_in_cam, debug, samples)
        prev_pose = np.asarray(prev_pose, np.float64)
        from .renderer import HipRenderer
        if self.one_call and isinstance(self.renderer, HipRenderer) and not self.renderer.full_frame and int(samples) <= 1:
            return self._on_track_one_call(prev_pose, current_rgb, current_depth)
        bb = U.compute_bbox(prev_pose, self.K, self.object_width, scale=(1000, 1000, 1000))
        dev = self._dev
        winA = (0, 0, self.image_size[0], self.image_size[0])
        if isinstance(self.renderer, HipRenderer) and self.renderer.full_frame:
            # pyrender route: the full rendered frame stays on the device and is cropped by the same kernel (and the
            # same bbox) as the camera frame -- predict.py:209-213 without the host round trip
            rgbA_d, depA_d = self.renderer.render_frame_device(prev_pose, self.K)
            winA = U.crop_window(bb)
        elif isinstance(self.renderer, HipRenderer):   # rendered A never leaves the device
            rgbA_d, depA_d = self.renderer.render_device(
                prev_pose, self.K, HipRenderer.gl_window(prev_pose, self.K, self.object_width))
        else:
            rgbA, depthA = self.render_window(prev_pose)
            rgbA_d = torch.from_numpy(np.ascontiguousarray(rgbA)).to(dev, non_blocking=True)
            depA_d = torch.from_numpy(np.ascontiguousarray(depthA).astype(np.uint16).view(np.int16)).to(dev, non_blocking=True)
        rgb_d = torch.from_numpy(np.ascontiguousarray(current_rgb)).to(dev, non_blocking=True)
        dep_d = torch.from_numpy(_depth_u16(current_depth, current_rgb)).to(dev, non_blocking=True)
        z_mm = float(prev_pose[2, 3]) * 1000
        # the reference evaluates `samples` IDENTICAL hypotheses (only i == 0 sets sample_pose, predict.py:229-231)
        # and returns the first: any count beyond the engine's batch capacity adds nothing -- clamp, never overrun
        n = max(1, min(int(samples), self.engine.max_batch))
        cropA = dict(rgb=rgbA_d, depth=depA_d, window=winA, z_offset_mm=z_mm, stats=0)
        cropB = dict(rgb=rgb_d, depth=dep_d, window=U.crop_window(bb), z_offset_mm=z_mm, stats=1)
        self.engine.preprocess([cropA] * n, self.engine.input_buffer_ptr(0))
        self.engine.preprocess([cropB] * n, self.engine.input_buffer_ptr(1))
        self._poseA[:n].copy_(torch.from_numpy(np.tile(prev_pose.reshape(1, 16), (n, 1))), non_blocking=True)
        self.engine.infer(self.engine.input_buffer_ptr(0), self.engine.input_buffer_ptr(1), n, NHWC,
                          self._trans, self._rot, self._poseA, self._poseB)
        poseB, trans_h, rot_h = self._read_back(n)               # one D2H + sync, as predict.py:275-276
        self.last_prediction = dict(trans=trans_h, rot=rot_h, bbox=bb)
        self.prev_rgb = current_rgb
        self.prev_depth = current_depth
        self.frame_cnt += 1
        return poseB[0]

    def _on_track_one_call(self, prev_pose, current_rgb, current_depth):
        """The whole frame in ONE library call (se3tn_on_track): compute_bbox, image A, both crops in one launch, network, pose
        update, read-back; the camera frame goes up as the window's rows / columns only, through pinned memory, together with the
        pose.  Same arithmetic as the step-by-step path below (tests/test_tracker_surface.py compares the two)."""
        rgb = current_rgb if (type(current_rgb) is np.ndarray and current_rgb.dtype == np.uint8 and current_rgb.flags.c_contiguous) \
            else np.ascontiguousarray(current_rgb, dtype=np.uint8)
        if rgb.ndim != 3 or rgb.shape[2] != 3:
            raise ValueError("rgb must be HxWx3 uint8")
        dep = current_depth if (type(current_depth) is np.ndarray and current_depth.dtype == np.uint16 and current_depth.flags.c_contiguous
                                and current_depth.shape == rgb.shape[:2]) else _depth_u16(current_depth, rgb)
        st = self._one_call_state
        if st is None:   # per-tracker constants of the call: argument objects are built once, not per frame
            import ctypes as C
            from ._lib import check
            from .engine import _stream_ptr
            st = self._one_call_state = dict(
                C=C, check=check, stream=_stream_ptr, fn=self.engine.lib.se3tn_on_track, P=np.empty((4, 4), np.float64),
                K=np.ascontiguousarray(self.K, np.float64), pose=np.empty((4, 4), np.float64), tr=np.empty(3, np.float32),
                ro=np.empty(3, np.float32), bb=np.empty((4, 2), np.int32))
            for k, t in (("P", C.c_double), ("K", C.c_double), ("pose", C.c_double), ("tr", C.c_float), ("ro", C.c_float), ("bb", C.c_int32)):
                st["p_" + k] = st[k].ctypes.data_as(C.POINTER(t))
        C = st["C"]
        st["P"][...] = prev_pose
        r = self.renderer
        st["check"](st["fn"](self.engine._h, r._m, st["p_P"], st["p_K"], C.c_double(float(self.object_width)), C.c_void_p(rgb.ctypes.data),
                             C.c_void_p(dep.ctypes.data), int(rgb.shape[0]), int(rgb.shape[1]), C.c_void_p(r.rgb.data_ptr()),
                             C.c_void_p(r.depth.data_ptr()), st["p_pose"], st["p_tr"], st["p_ro"], st["p_bb"], st["stream"]()),
                    "se3tn_on_track")
        self.last_prediction = dict(trans=st["tr"].reshape(1, 3).copy(), rot=st["ro"].reshape(1, 3).copy(), bbox=st["bb"].copy())
        self.prev_rgb = current_rgb
        self.prev_depth = current_depth
        self.frame_cnt += 1
        return st["pose"].copy()

    def on_track_batch(self, prev_poses, rgbs, depths):
        """Extension: n independent (pose, frame) pairs of the SAME object in one engine call -- several
        sequences / cameras / hypotheses (frames of one track are serial, so this is where batch > 1
        comes from).  Same arithmetic per pair as on_track; returns [n,4,4] float64.
        With the built-in rasteriser the whole step is ONE library call (se3tn_on_track_batch: image A of all n poses in four
        launches, the frames' crop windows staged through pinned memory in one copy, no per-pair Python)."""
        from .renderer import HipRenderer
        n = len(prev_poses)
        if n > self.engine.max_batch:
            raise ValueError("on_track_batch: %d pairs > max_samples=%d given to Tracker()" % (n, self.engine.max_batch))
        if self.one_call and isinstance(self.renderer, HipRenderer) and not self.renderer.full_frame and n > 0:
            return self._on_track_batch_one_call(prev_poses, rgbs, depths)
        return self._on_track_batch_stepwise(prev_poses, rgbs, depths)

    def _on_track_batch_one_call(self, prev_poses, rgbs, depths):
        import ctypes as C
        from ._lib import check
        from .engine import _stream_ptr
        n = len(prev_poses)
        poses = np.ascontiguousarray(np.stack([np.asarray(p, np.float64) for p in prev_poses]).reshape(n, 16))
        frames_rgb, frames_dep = [], []
        for i in range(n):
            rgb = rgbs[i]
            if not (type(rgb) is np.ndarray and rgb.dtype == np.uint8 and rgb.flags.c_contiguous):
                rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            if rgb.ndim != 3 or rgb.shape[2] != 3:
                raise ValueError("rgb must be HxWx3 uint8")
            dep = depths[i]
            if not (type(dep) is np.ndarray and dep.dtype == np.uint16 and dep.flags.c_contiguous and dep.shape == rgb.shape[:2]):
                dep = _depth_u16(dep, rgb)
            if rgb.shape != frames_rgb[0].shape if frames_rgb else False:
                raise ValueError("on_track_batch: the frames of one call must have one size")
            frames_rgb.append(rgb); frames_dep.append(dep)
        H, W = int(frames_rgb[0].shape[0]), int(frames_rgb[0].shape[1])
        st = self._batch_state
        if st is None or st["n"] < n:
            dev = self._dev
            st = self._batch_state = dict(
                n=n, rgbA=torch.empty((n, 176, 176, 3), dtype=torch.uint8, device=dev),
                depthA=torch.empty((n, 176, 176), dtype=torch.int16, device=dev),
                K=np.ascontiguousarray(self.K, np.float64))
        out = np.empty((n, 16), np.float64)
        tr, ro, bb = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32), np.empty((n, 4, 2), np.int32)
        prgb = (C.c_void_p * n)(*[f.ctypes.data for f in frames_rgb])
        pdep = (C.c_void_p * n)(*[f.ctypes.data for f in frames_dep])
        check(self.engine.lib.se3tn_on_track_batch(
            self.engine._h, self.renderer._m, n, C.c_void_p(poses.ctypes.data), st["K"].ctypes.data_as(C.POINTER(C.c_double)),
            C.c_double(float(self.object_width)), prgb, pdep, H, W, C.c_void_p(st["rgbA"].data_ptr()), C.c_void_p(st["depthA"].data_ptr()),
            C.c_void_p(out.ctypes.data), C.c_void_p(tr.ctypes.data), C.c_void_p(ro.ctypes.data), C.c_void_p(bb.ctypes.data), _stream_ptr()),
            "se3tn_on_track_batch")
        self.last_prediction = dict(trans=tr, rot=ro, bbox=bb, rgbA=list(st["rgbA"][:n]), depthA=list(st["depthA"][:n]))
        self.frame_cnt += 1
        return out.reshape(n, 4, 4)

    def _on_track_batch_stepwise(self, prev_poses, rgbs, depths):
        """on_track_batch step by step (injected renderers, the pyrender route, one_call = False): per pair a render and two uploads"""
        from .renderer import HipRenderer
        n = len(prev_poses)
        dev = self._dev
        cropsA, cropsB, keep, bboxes = [], [], [], []
        poses = np.stack([np.asarray(p, np.float64) for p in prev_poses])
        for i in range(n):
            bb = U.compute_bbox(poses[i], self.K, self.object_width, scale=(1000, 1000, 1000))
            bboxes.append(bb)
            if isinstance(self.renderer, HipRenderer) and not self.renderer.full_frame:
                rgbA_d = torch.empty((176, 176, 3), dtype=torch.uint8, device=dev)
                depA_d = torch.empty((176, 176), dtype=torch.int16, device=dev)
                self.renderer.render_device(poses[i], self.K, HipRenderer.gl_window(poses[i], self.K, self.object_width),
                                            rgbA_d, depA_d)
            else:
                rgbA, depthA = self.render_window(poses[i])
                rgbA_d = torch.from_numpy(np.ascontiguousarray(rgbA)).to(dev)
                depA_d = torch.from_numpy(np.ascontiguousarray(depthA).astype(np.uint16).view(np.int16)).to(dev)
            rgb_d = torch.from_numpy(np.ascontiguousarray(rgbs[i])).to(dev, non_blocking=True)
            dep_d = torch.from_numpy(_depth_u16(depths[i], rgbs[i])).to(dev, non_blocking=True)
            keep += [rgbA_d, depA_d, rgb_d, dep_d]
            z_mm = float(poses[i, 2, 3]) * 1000
            cropsA.append(dict(rgb=rgbA_d, depth=depA_d, window=(0, 0, 176, 176), z_offset_mm=z_mm, stats=0))
            cropsB.append(dict(rgb=rgb_d, depth=dep_d, window=U.crop_window(bb), z_offset_mm=z_mm, stats=1))
        self.engine.preprocess(cropsA, self.engine.input_buffer_ptr(0))
        self.engine.preprocess(cropsB, self.engine.input_buffer_ptr(1))
        self._poseA[:n].copy_(torch.from_numpy(poses.reshape(n, 16)), non_blocking=True)
        self.engine.infer(self.engine.input_buffer_ptr(0), self.engine.input_buffer_ptr(1), n, NHWC,
                          self._trans, self._rot, self._poseA, self._poseB)
        out, trans_h, rot_h = self._read_back(n)
        # what on_track keeps in last_prediction / renderer.rgb, per pair (callers that log or check the step)
        self.last_prediction = dict(trans=trans_h, rot=rot_h, bbox=np.stack(bboxes), rgbA=keep[0::4], depthA=keep[1::4])
        self.frame_cnt += 1
        return out
