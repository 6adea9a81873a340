"""MI355X-native (gfx950) implementation of the se(3)-TrackNet per-frame inference hot path
(reference: wenbowen123/iros20-6d-pose-tracking, predict.py `Tracker.on_track`).

Public surface mirrors the reference's for this path: ``Tracker`` (predict.py:127),
``Se3TrackNet`` (se3_tracknet.py:52), ``compute_bbox`` (Utils.py:302).  All arithmetic is in
libse3tracknet.so (hand-written HIP, C ABI in include/se3tracknet.h); importing this package
without the built library raises ImportError -- there is no fallback path."""
from . import _lib
from .engine import Engine, PipelinedEngine, NCHW, NHWC, compute_bbox as compute_bbox_c, pack_crops, pose_update_host
from .se3_tracknet import Se3TrackNet
from .tracker import Tracker
from .utils import compute_bbox, crop_window
from . import metrics, sequence
from .renderer import HipRenderer
from .live import LiveTracker, quaternion_from_matrix

_lib.load()

__all__ = ["Engine", "PipelinedEngine", "Se3TrackNet", "Tracker", "compute_bbox", "crop_window", "pack_crops", "pose_update_host", "NCHW", "NHWC",
           "metrics", "sequence", "HipRenderer", "LiveTracker", "quaternion_from_matrix"]
