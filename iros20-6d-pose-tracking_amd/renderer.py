"""HIP rasteriser front end: drop-in for the reference's VispyRenderer as used by
Tracker.render_window (vispy_renderer.py:47-178, predict.py:193-215) -- no OpenGL.
The rendered rgbA / depthA stay on the device and feed se3tn_preprocess directly."""
import ctypes as C

import numpy as np
import torch

from . import utils as U
from ._lib import check
from .engine import _stream_ptr


class HipRenderer:
    def __init__(self, engine, model, resolution=176, mode="vispy", frame_size=None):
        """model: path to a .ply with faces / vertex colours / normals (or a textured .obj), or a dict with
        vertices [V,3], faces [F,3], colors [V,3] (0..255), normals [V,3] (optional), and for mode 'pyrender'
        optionally uv [V,2], texture uint8 [h,w,3], kd [3].
        mode 'vispy': the reference's VispyRenderer (176 x 176 window, Lambert shader); mode 'pyrender': its
        offscreen_renderer.Renderer (full camera frame of frame_size = (H, W), ambient light only), selected by
        dataset_info['renderer'] == 'pyrenderer' (predict.py:161-164)."""
        assert resolution == 176 and mode in ("vispy", "pyrender")
        self.engine = engine
        self.mode = mode
        self.full_frame = mode == "pyrender"
        if isinstance(model, str):
            mesh = U.load_obj_mesh(model) if model.endswith(".obj") else U.load_ply_mesh(model)
        else:
            mesh = model
        v = np.ascontiguousarray(mesh["vertices"], np.float32)
        f = np.ascontiguousarray(mesh["faces"], np.int32)
        col = np.ascontiguousarray(np.asarray(mesh["colors"], np.float64) / 255.0, np.float32)
        nrm = mesh.get("normals")
        if nrm is None:
            nrm = U.vertex_normals(v, f)
        nrm = np.asarray(nrm)
        if nrm.dtype != np.float32:                                        # a float32 .ply property is normalised in float32 (:126)
            nrm = nrm.astype(np.float64)
        nrm = np.ascontiguousarray(nrm / np.linalg.norm(nrm, axis=1).reshape(-1, 1), np.float32)
        self.mesh = dict(vertices=v, faces=f, colors01=col, normals=nrm)
        h = C.c_void_p()
        check(engine.lib.se3tn_mesh_create(engine._h, v.ctypes.data, nrm.ctypes.data, col.ctypes.data, len(v),
                                           f.ctypes.data, len(f), C.byref(h)), "se3tn_mesh_create")
        self._m = h
        dev = "cuda:%d" % engine.device
        if self.full_frame:
            assert frame_size is not None, "mode 'pyrender' renders the whole camera frame: pass frame_size=(H, W)"
            uv, texture = mesh.get("uv"), mesh.get("texture")
            kd = (C.c_float * 3)(*[float(x) for x in mesh.get("kd", (1.0, 1.0, 1.0))])
            if texture is not None and uv is not None:
                uv32 = np.ascontiguousarray(uv, np.float32)
                tex = np.ascontiguousarray(texture, np.uint8)
                check(engine.lib.se3tn_mesh_set_texture(self._m, uv32.ctypes.data, tex.ctypes.data, tex.shape[1], tex.shape[0], kd),
                      "se3tn_mesh_set_texture")
            else:
                check(engine.lib.se3tn_mesh_set_texture(self._m, None, None, 0, 0, kd), "se3tn_mesh_set_texture")
            self.H, self.W = int(frame_size[0]), int(frame_size[1])
            self.rgb = torch.empty((self.H, self.W, 3), dtype=torch.uint8, device=dev)
            self.depth = torch.empty((self.H, self.W), dtype=torch.int16, device=dev)
            return
        self.rgb = torch.empty((176, 176, 3), dtype=torch.uint8, device=dev)
        self.depth = torch.empty((176, 176), dtype=torch.int16, device=dev)  # uint16 bits

    def __del__(self):
        try:
            if getattr(self, "_m", None):
                self.engine.lib.se3tn_mesh_destroy(self._m)
                self._m = None
        except Exception:
            pass

    @staticmethod
    def gl_window(ob2cam, K, object_width):
        """left, top, right, bottom as Tracker.render_window passes them to update_cam_mat
        (predict.py:201-207): compute_bbox in the y-flipped GL image."""
        bbox = U.compute_bbox(np.asarray(ob2cam, np.float64), K, object_width, scale=(1000, -1000, 1000))
        return (int(np.min(bbox[:, 1])), int(np.min(bbox[:, 0])), int(np.max(bbox[:, 1])), int(np.max(bbox[:, 0])))

    def render_device(self, ob2cam, K, gl_window, rgb=None, depth=None):
        """Renders into device tensors (uint8 [176,176,3], uint16-as-int16 [176,176]); asynchronous."""
        rgb = self.rgb if rgb is None else rgb
        depth = self.depth if depth is None else depth
        p = (C.c_double * 16)(*np.asarray(ob2cam, np.float64).reshape(16))
        k = (C.c_double * 9)(*np.asarray(K, np.float64).reshape(9))
        w = (C.c_int32 * 4)(*[int(x) for x in gl_window])
        check(self.engine.lib.se3tn_render(self.engine._h, self._m, p, k, w, C.c_void_p(rgb.data_ptr()),
                                           C.c_void_p(depth.data_ptr()), _stream_ptr()), "se3tn_render")
        return rgb, depth

    def render_frame_device(self, ob2cam, K):
        """mode 'pyrender': the full camera frame, on the device (uint8 [H,W,3], uint16-as-int16 [H,W] mm); asynchronous."""
        assert self.full_frame
        p = (C.c_double * 16)(*np.asarray(ob2cam, np.float64).reshape(16))
        k = (C.c_double * 9)(*np.asarray(K, np.float64).reshape(9))
        check(self.engine.lib.se3tn_render_frame(self.engine._h, self._m, p, k, self.W, self.H, C.c_void_p(self.rgb.data_ptr()),
                                                 C.c_void_p(self.depth.data_ptr()), _stream_ptr()), "se3tn_render_frame")
        return self.rgb, self.depth

    def render_frame(self, ob2cam, K):
        """numpy (rgb uint8 [H,W,3], depth uint16 [H,W] mm) == (color, (depth * 1000).astype(uint16)) of
        offscreen_renderer.Renderer.render([ob2cam]) (predict.py:210-211)."""
        rgb, depth = self.render_frame_device(ob2cam, K)
        return rgb.cpu().numpy(), depth.cpu().numpy().view(np.uint16)

    def render(self, ob2cam, K, gl_window):
        """numpy (rgb uint8 [176,176,3], depth uint16 [176,176] mm), like VispyRenderer.render_image."""
        rgb, depth = self.render_device(ob2cam, K, gl_window)
        return rgb.cpu().numpy(), depth.cpu().numpy().view(np.uint16)
