"""Thin Python owner of one se3tn context (one per process / GPU).  PyTorch is used only for
device memory, streams and (in dist.py) torch.distributed -- all arithmetic is in the HIP library."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import CROP_DTYPE, NCHW, NHWC, RES, Crop, Se3tnError, check


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_crops(rgb, depth, windows, z_offset_mm, stats):
    """Vectorised se3tn_crop descriptors for a batch of frames that live in ONE tensor each:
    rgb cuda u8 [n,H,W,3], depth cuda 2-byte [n,H,W], windows int [n,4] (left,top,right,bottom),
    z_offset_mm float [n], stats 0|1.  Returns a numpy record array accepted by Engine.preprocess."""
    assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.is_contiguous() and rgb.dim() == 4 and rgb.shape[3] == 3
    assert depth.is_cuda and depth.element_size() == 2 and depth.is_contiguous() and depth.shape == rgb.shape[:3]
    n, H, W = int(rgb.shape[0]), int(rgb.shape[1]), int(rgb.shape[2])
    rec = np.zeros(n, dtype=CROP_DTYPE)
    idx = np.arange(n, dtype=np.uint64)
    rec["rgb"] = np.uint64(rgb.data_ptr()) + idx * np.uint64(H * W * 3)
    rec["depth"] = np.uint64(depth.data_ptr()) + idx * np.uint64(H * W * 2)
    rec["H"] = H; rec["W"] = W
    win = np.asarray(windows, dtype=np.int64).reshape(n, 4)
    rec["left"], rec["top"], rec["right"], rec["bottom"] = win[:, 0], win[:, 1], win[:, 2], win[:, 3]
    rec["z_offset_mm"] = np.asarray(z_offset_mm, dtype=np.float64)
    rec["stats"] = int(stats)
    return rec


class Engine:
    def __init__(self, device=0, max_batch=64):
        self.lib = _lib.load()
        self.device = int(device)
        self.max_batch = int(max_batch)
        h = C.c_void_p()
        check(self.lib.se3tn_create(self.device, self.max_batch, C.byref(h)), "se3tn_create")
        self._h = h
        self._blob = None  # keeps a bound (caller-owned) weight blob alive
        self.has_weights = False
        if self.device >= 0:
            torch.cuda.set_device(self.device)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.se3tn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, H, W):
        """Start-up reservation (se3tn_reserve): z-buffer / fill_depth scratch for H x W camera frames and the planes of the
        selected modes, so that no stream-ordered call allocates later (hipGraph capture, latency of the first frame)."""
        check(self.lib.se3tn_reserve(self._h, int(H), int(W)), "se3tn_reserve")

    def split_weights_device(self):
        """f16x3 mode: copy of the device-derived split panels as a CPU uint8 tensor (tests)."""
        n = int(self.lib.se3tn_split_weights_bytes(self._h))
        ptr = self.lib.se3tn_split_weights_device(self._h)
        if not ptr:
            raise Se3tnError("split panels not derived: select PREC_F16X3 with weights bound")
        out = torch.empty(n, dtype=torch.uint8, device="cuda:%d" % self.device)
        check(self.lib.se3tn_memcpy_d2d(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), n, _stream_ptr()), "se3tn_memcpy_d2d")
        return out.cpu()

    def split_weights_host(self, packed_blob):
        """The host statement of the same derivation applied to a packed blob (CPU uint8 tensor)."""
        n = int(self.lib.se3tn_split_weights_bytes(self._h))
        blob = packed_blob.contiguous()
        out = torch.empty(n, dtype=torch.uint8)
        check(self.lib.se3tn_split_weights_host(C.c_void_p(blob.data_ptr()), blob.numel(), C.c_void_p(out.data_ptr()), n),
              "se3tn_split_weights_host")
        return out

    # ---- weights ---------------------------------------------------------------------------
    def pack_state_dict(self, state_dict):
        """Hand the reference checkpoint's state_dict (predict.py:151-156) to the library:
        BN folding + packing happen in C++.  Returns the packed blob as a CPU uint8 tensor."""
        n = 0
        for key, t in state_dict.items():
            if t.dtype != torch.float32:
                if key.endswith("num_batches_tracked"):
                    continue
                raise Se3tnError("state_dict[%s]: expected float32, got %s" % (key, t.dtype))
            t = t.detach().cpu().contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            check(self.lib.se3tn_set_tensor(self._h, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim()),
                  "se3tn_set_tensor(%s)" % key)
            n += 1
        check(self.lib.se3tn_pack_weights(self._h), "se3tn_pack_weights")
        nbytes = self.lib.se3tn_packed_bytes(self._h)
        host = self.lib.se3tn_packed_host(self._h)
        buf = (C.c_uint8 * nbytes).from_address(host)
        return torch.frombuffer(buf, dtype=torch.uint8).clone()

    def load_state_dict(self, state_dict):
        self.pack_state_dict(state_dict)
        check(self.lib.se3tn_upload_weights(self._h, _stream_ptr()), "se3tn_upload_weights")
        self.has_weights = True

    def packed_bytes(self):
        return int(self.lib.se3tn_packed_bytes(self._h))

    def bind_blob(self, blob_cuda):
        """Use a caller-owned CUDA uint8 tensor holding the packed blob (e.g. received through
        an RCCL broadcast)."""
        assert blob_cuda.is_cuda and blob_cuda.dtype == torch.uint8 and blob_cuda.is_contiguous()
        torch.cuda.current_stream().synchronize()
        check(self.lib.se3tn_bind_weights(self._h, C.c_void_p(blob_cuda.data_ptr()), blob_cuda.numel()),
              "se3tn_bind_weights")
        self._blob = blob_cuda
        self.has_weights = True

    # ---- constants -------------------------------------------------------------------------
    def set_normalization(self, mean, std):
        m = (C.c_double * 8)(*[float(x) for x in np.asarray(mean).reshape(8)])
        s = (C.c_double * 8)(*[float(x) for x in np.asarray(std).reshape(8)])
        check(self.lib.se3tn_set_normalization(self._h, m, s), "se3tn_set_normalization")

    def enable_graphs(self, on=True):
        """hipGraph replay of repeated se3tn_infer calls (needs a non-default stream)."""
        check(self.lib.se3tn_enable_graphs(self._h, 1 if on else 0), "se3tn_enable_graphs")

    def set_precision(self, mode):
        """_lib.PREC_F32 (default, exact) or _lib.PREC_F16X3 (split-f16 MFMA for the deep layers, n >= 32)."""
        check(self.lib.se3tn_set_precision(self._h, int(mode)), "se3tn_set_precision")

    def set_offset_rule(self, rule):
        """Rounding of OffsetDepth's `depth -= pose[2,3]*1000`: "numpy1" | _lib.OFFSET_RULE_NUMPY1 (default: one float32 operation, what
        every NumPy the reference runs on does) or "numpy2" | _lib.OFFSET_RULE_NUMPY2 (float64, rounded once)."""
        rule = {"numpy1": _lib.OFFSET_RULE_NUMPY1, "numpy2": _lib.OFFSET_RULE_NUMPY2}.get(rule, rule)
        check(self.lib.se3tn_set_offset_rule(self._h, int(rule)), "se3tn_set_offset_rule")

    def get_offset_rule(self):
        return "numpy2" if self.lib.se3tn_get_offset_rule(self._h) == _lib.OFFSET_RULE_NUMPY2 else "numpy1"

    def set_small_kernels(self, on=True):
        """Batches of 1-5 pairs through the batch-1 kernel family (conv64_small, conv_slices_small; the small-tile stem + pool at 1-2
        pairs) -- the default -- or through the general kernels (False): include/se3tracknet.h, se3tn_set_small_kernels."""
        check(self.lib.se3tn_set_small_kernels(self._h, 1 if on else 0), "se3tn_set_small_kernels")

    def get_small_kernels(self):
        return bool(self.lib.se3tn_get_small_kernels(self._h))

    def set_raster_rule(self, sub_bits):
        """Sub-pixel bits of the rasteriser's window coordinates: 4 (default: the software GL the goldens were rendered on, = the GL
        minimum) or 8 (what desktop GPUs report for GL_SUBPIXEL_BITS).  include/se3tracknet.h: se3tn_set_raster_rule."""
        check(self.lib.se3tn_set_raster_rule(self._h, int(sub_bits)), "se3tn_set_raster_rule")

    def get_raster_rule(self):
        return int(self.lib.se3tn_get_raster_rule(self._h))

    def set_winograd(self, min_batch, tile=0):
        """Batches of n >= min_batch run the 256/512-channel residual blocks as Winograd F(tile x tile,3x3) launch sequences (float32;
        tile 2 | 4 | 6 | _lib.WINOGRAD_TILE_6_4 = 6 for the 256-channel block + 4 for the heads | _lib.WINOGRAD_TILE_AUTO = 4 below 14
        pairs, from there 6 for the 256-channel block and -- while rot_normalizer <= 0.2 rad -- for the heads; 0 = keep the tile);
        min_batch 0 = always the direct kernels.  Defaults: SE3TN_WINOGRAD_DEFAULT_MIN_BATCH / _TILE of include/se3tracknet.h."""
        check(self.lib.se3tn_set_winograd(self._h, int(min_batch), int(tile)), "se3tn_set_winograd")

    def set_trunk_winograd(self, min_batch, min_fill_percent=None):
        """Launches of the 64-channel trunk at n >= min_batch take the fused Winograd F(2x2,3x3) kernel when their workgroups fill
        whole rounds of the CUs to >= min_fill_percent (None = SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_FILL, 0 = every such launch);
        min_batch 0 = always the direct kernels."""
        fill = _lib.TRUNK_WINOGRAD_DEFAULT_MIN_FILL if min_fill_percent is None else int(min_fill_percent)
        check(self.lib.se3tn_set_trunk_winograd(self._h, int(min_batch), fill), "se3tn_set_trunk_winograd")

    def get_trunk_winograd(self):
        """(min_batch, min_fill_percent) currently in force."""
        mb, fp = C.c_int(), C.c_int()
        check(self.lib.se3tn_get_trunk_winograd(self._h, C.byref(mb), C.byref(fp)), "se3tn_get_trunk_winograd")
        return mb.value, fp.value

    def get_winograd(self):
        """(min_batch, tile) currently in force."""
        mb, t = C.c_int(), C.c_int()
        check(self.lib.se3tn_get_winograd(self._h, C.byref(mb), C.byref(t)), "se3tn_get_winograd")
        return mb.value, t.value

    def keep_intermediates(self, on=True):
        """Make the fused Winograd blocks also store the activations they otherwise keep on chip ("ab_t", "head_t",
        "head" of debug_buffer); results are bit-identical either way -- except at 1-5 pairs, where the stem and the tail then take the
        general kernels that write "stem" / "head" (same tolerances, another summation order)."""
        check(self.lib.se3tn_keep_intermediates(self._h, 1 if on else 0), "se3tn_keep_intermediates")

    def overflow(self):
        """True if a split-row store left the f16 range since the last call (synchronises)."""
        f = C.c_int(0)
        check(self.lib.se3tn_overflow(self._h, C.byref(f)), "se3tn_overflow")
        return bool(f.value)

    def set_normalizers(self, trans_normalizer, rot_normalizer):
        check(self.lib.se3tn_set_normalizers(self._h, float(trans_normalizer), float(rot_normalizer)),
              "se3tn_set_normalizers")

    # ---- compute ---------------------------------------------------------------------------
    def input_buffer_ptr(self, which):
        return self.lib.se3tn_input_buffer(self._h, which)

    def preprocess(self, crops, out):
        """crops: list of dict(rgb=cuda u8 [H,W,3], depth=cuda u16-as-int16/uint16 [H,W],
        window=(left,top,right,bottom), z_offset_mm=float, stats=0|1), or the packed descriptor array
        returned by pack_crops() (no per-crop Python work).
        out: cuda float32 tensor [n,176,176,4] or a raw device pointer (int)."""
        if isinstance(crops, np.ndarray):
            assert crops.dtype == CROP_DTYPE and crops.flags.c_contiguous
            n = crops.shape[0]
            arr = crops.ctypes.data_as(C.POINTER(Crop))
        else:
            n = len(crops)
            arr = (Crop * n)()
            for i, c in enumerate(crops):
                rgb, depth = c["rgb"], c["depth"]
                assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.is_contiguous() and rgb.shape[2] == 3
                assert depth.is_cuda and depth.element_size() == 2 and depth.is_contiguous()
                arr[i].rgb = rgb.data_ptr(); arr[i].depth = depth.data_ptr()
                arr[i].H, arr[i].W = int(rgb.shape[0]), int(rgb.shape[1])
                arr[i].left, arr[i].top, arr[i].right, arr[i].bottom = [int(v) for v in c["window"]]
                arr[i].z_offset_mm = float(c["z_offset_mm"])
                arr[i].stats = int(c["stats"])
        ptr = out.data_ptr() if torch.is_tensor(out) else int(out)
        check(self.lib.se3tn_preprocess(self._h, arr, n, C.c_void_p(ptr), _stream_ptr()), "se3tn_preprocess")

    def crop_raw(self, rgb, depth, window):
        """crop_bbox (Utils.py:320-359) alone: numpy rgb u8 [H,W,3] + depth u16 [H,W], window (left, top, right,
        bottom) -> numpy (rgb u8 [176,176,3], depth u16 [176,176])."""
        dev = "cuda:%d" % self.device
        rgb_d = rgb if torch.is_tensor(rgb) else torch.from_numpy(np.ascontiguousarray(rgb, dtype=np.uint8)).to(dev)
        dep_d = depth if torch.is_tensor(depth) else torch.from_numpy(np.ascontiguousarray(depth, dtype=np.uint16).view(np.int16)).to(dev)
        assert rgb_d.is_cuda and dep_d.is_cuda and rgb_d.dtype == torch.uint8 and dep_d.element_size() == 2
        assert rgb_d.dim() == 3 and rgb_d.shape[2] == 3 and tuple(dep_d.shape) == tuple(rgb_d.shape[:2])
        c = Crop()
        c.rgb = rgb_d.data_ptr(); c.depth = dep_d.data_ptr()
        c.H, c.W = int(rgb_d.shape[0]), int(rgb_d.shape[1])
        c.left, c.top, c.right, c.bottom = [int(v) for v in window]
        out_rgb = torch.empty((RES, RES, 3), dtype=torch.uint8, device=dev)
        out_d = torch.empty((RES, RES), dtype=torch.int16, device=dev)
        check(self.lib.se3tn_crop_raw(self._h, C.byref(c), C.c_void_p(out_rgb.data_ptr()), C.c_void_p(out_d.data_ptr()),
                                      _stream_ptr()), "se3tn_crop_raw")
        return out_rgb.cpu().numpy(), out_d.cpu().numpy().view(np.uint16)

    def fill_depth(self, depth_mm, max_depth=2.0, extrapolate=False, blur_type="bilateral", return_metres=False):
        """Utils.py:455-514 fill_depth as predict_ros.py:38-41 applies it: uint16 millimetre frame (numpy [H,W] or a
        cuda int16/uint16 tensor) -> hole-filled uint16 millimetres (same container kind); return_metres=True also
        returns the float32 metre image fill_depth itself returns."""
        dev = "cuda:%d" % self.device
        is_np = not torch.is_tensor(depth_mm)
        d = torch.from_numpy(np.ascontiguousarray(depth_mm, dtype=np.uint16).view(np.int16)).to(dev) if is_np else depth_mm
        assert d.is_cuda and d.element_size() == 2 and d.dim() == 2 and d.is_contiguous()
        H, W = int(d.shape[0]), int(d.shape[1])
        out = torch.empty((H, W), dtype=torch.int16, device=dev)
        out_m = torch.empty((H, W), dtype=torch.float32, device=dev) if return_metres else None
        blur = {"bilateral": _lib.BLUR_BILATERAL, "gaussian": _lib.BLUR_GAUSSIAN}.get(blur_type, _lib.BLUR_NONE)
        check(self.lib.se3tn_fill_depth(self._h, C.c_void_p(d.data_ptr()), H, W, float(max_depth), 1 if extrapolate else 0, blur,
                                        C.c_void_p(out.data_ptr()), C.c_void_p(out_m.data_ptr()) if return_metres else None,
                                        _stream_ptr()), "se3tn_fill_depth")
        if is_np:
            mm = out.cpu().numpy().view(np.uint16)
            return (mm, out_m.cpu().numpy()) if return_metres else mm
        return (out, out_m) if return_metres else out

    def infer(self, A, B, n, layout=NCHW, trans=None, rot=None, poseA=None, poseB=None):
        def p(x):
            if x is None:
                return None
            return C.c_void_p(x.data_ptr() if torch.is_tensor(x) else int(x))
        check(self.lib.se3tn_infer(self._h, p(A), p(B), int(n), int(layout), p(trans), p(rot), p(poseA), p(poseB),
                                   _stream_ptr()), "se3tn_infer")

    def feature(self, n):
        out = torch.empty((n, 256, 22, 22), dtype=torch.float32, device="cuda:%d" % self.device)
        check(self.lib.se3tn_get_feature(self._h, n, C.c_void_p(out.data_ptr()), _stream_ptr()), "se3tn_get_feature")
        return out

    def logits(self, n):
        out = torch.empty((n, 6), dtype=torch.float32, device="cuda:%d" % self.device)
        check(self.lib.se3tn_memcpy_d2d(C.c_void_p(out.data_ptr()), C.c_void_p(self.lib.se3tn_logits(self._h)),
                                        n * 6 * 4, _stream_ptr()), "se3tn_memcpy_d2d")
        return out

    def debug_buffer(self, name, n):
        """Copy of an internal NHWC activation buffer: float32 cuda tensor [n,H,W,C]."""
        ptr = C.c_void_p()
        dims = (C.c_int32 * 3)()
        check(self.lib.se3tn_debug_buffer(self._h, name.encode(), C.byref(ptr), dims), "se3tn_debug_buffer")
        out = torch.empty((n, dims[0], dims[1], dims[2]), dtype=torch.float32, device="cuda:%d" % self.device)
        check(self.lib.se3tn_memcpy_d2d(C.c_void_p(out.data_ptr()), ptr, out.numel() * 4, _stream_ptr()),
              "se3tn_memcpy_d2d")
        return out

    # ---- profiling -------------------------------------------------------------------------
    def profile_enable(self, slots=1):
        check(self.lib.se3tn_profile_enable(self._h, int(slots)), "se3tn_profile_enable")

    def profile_read(self, slot=0):
        """(ms inside the conv3x3 MFMA family, number of such launches, ms of all launches)"""
        conv = C.c_float(); nl = C.c_int(); tot = C.c_float()
        check(self.lib.se3tn_profile_read(self._h, slot, C.byref(conv), C.byref(nl), C.byref(tot)),
              "se3tn_profile_read")
        return conv.value, nl.value, tot.value

    def profile_launches(self, slot=0):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        k = self.lib.se3tn_profile_launches(self._h, slot, 32, names, ms)
        if k <= 0:
            raise Se3tnError("se3tn_profile_launches: " + self.lib.se3tn_last_error().decode())
        return [(names[i].decode(), ms[i]) for i in range(k)]


class PipelinedEngine:
    """Throughput mode: `depth` contexts (activation workspaces) that share ONE packed weight blob, each with its own HIP
    stream; successive batches go to successive lanes, so the HBM-bound passes of one batch (crop / normalise, max-pool,
    Winograd transforms, pose) run under the matrix-core-bound kernels of the other.  Measured at batch 64: two lanes
    35.7 k pairs/s vs 33.0 k on one stream; a third lane adds nothing (scripts/two_stream_probe.py).

        pe = PipelinedEngine(0, 64); pe.load_state_dict(sd); pe.set_normalization(mean, std)
        for batch in batches:
            eng, stream = pe.next_lane()
            with torch.cuda.stream(stream):
                eng.preprocess(...); eng.infer(...)          # asynchronous; outputs must be per-lane buffers
        pe.synchronize()
    """

    def __init__(self, device=0, max_batch=64, depth=2):
        assert depth >= 1
        self.device = int(device)
        self.engines = [Engine(device, max_batch) for _ in range(depth)]
        dev = "cuda:%d" % self.device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self._i = 0
        self._blob = None

    def load_state_dict(self, state_dict):
        """Fold + pack once, upload once, bind the same device blob to every lane."""
        blob = self.engines[0].pack_state_dict(state_dict).to("cuda:%d" % self.device)
        self.bind_blob(blob)

    def bind_blob(self, blob_cuda):
        self._blob = blob_cuda
        for e in self.engines:
            e.bind_blob(blob_cuda)

    def __getattr__(self, name):
        # configuration calls (set_normalization, set_normalizers, set_precision, set_winograd, ...) go to every lane
        if name.startswith("set_") or name in ("keep_intermediates", "enable_graphs"):
            def fan_out(*a, **k):
                for e in self.engines:
                    getattr(e, name)(*a, **k)
            return fan_out
        raise AttributeError(name)

    def next_lane(self):
        k = self._i % len(self.engines)
        self._i += 1
        return self.engines[k], self.streams[k]

    def synchronize(self):
        for s in self.streams:
            s.synchronize()


# ---- host-side float64 helpers (pure CPU entry points of the C ABI) ---------------------------
def compute_bbox(pose, K, object_width_mm):
    """Utils.py:302-316 with scale=(1000,1000,1000): int32 [4,2] (v,u)."""
    lib = _lib.load()
    p = (C.c_double * 16)(*np.asarray(pose, np.float64).reshape(16))
    k = (C.c_double * 9)(*np.asarray(K, np.float64).reshape(9))
    out = (C.c_int32 * 8)()
    check(lib.se3tn_compute_bbox(p, k, float(object_width_mm), out), "se3tn_compute_bbox")
    return np.array(out[:], dtype=np.int32).reshape(4, 2)


def pose_update_host(A_in_cam, trans, rot, trans_normalizer, rot_normalizer):
    """datasets.py:159-175 processPredict (host float64)."""
    lib = _lib.load()
    a = (C.c_double * 16)(*np.asarray(A_in_cam, np.float64).reshape(16))
    t = (C.c_float * 3)(*np.asarray(trans, np.float32).reshape(3))
    r = (C.c_float * 3)(*np.asarray(rot, np.float32).reshape(3))
    out = (C.c_double * 16)()
    check(lib.se3tn_pose_update_host(a, t, r, float(trans_normalizer), float(rot_normalizer), out),
          "se3tn_pose_update_host")
    return np.array(out[:], dtype=np.float64).reshape(4, 4)
