"""Inference-time stand-in for the reference's ``TrackDataset`` as ``Tracker.dataset`` holds it
(predict.py:189-191: mode 'eval', posttransforms OffsetDepth -> NormalizeChannels -> ToTensor): the two
calls on the per-frame path,

    sample = tracker.dataset.processData(rgbA, depthA, A_in_cam, rgbB, depthB, B_in_cam)[0]   predict.py:264
    pose   = tracker.dataset.processPredict(A_in_cam, (trans, rot))                            predict.py:277

with the reference's argument meaning and return structure (datasets.py:115-175).  The arithmetic runs in
the HIP library: processData = se3tn_preprocess on the two already-cropped 176x176 image pairs (identity
window; same kernel, same float64 rules as the fused path Tracker.on_track uses), processPredict =
se3tn_pose_update_host.  ``Tracker.on_track`` itself does not go through this object (it never leaves the
device between the crop and the pose); the shim is the reference's inner boundary for callers that drive
the stages themselves."""
import numpy as np
import torch

from .engine import pose_update_host


def rotation_matrix_to_rotvec(R):
    """cv2.Rodrigues(3x3) -> rotation vector: the label math of datasets.py:148 (unused at inference, kept
    so that processData returns the reference's tuple).  Float64; the matrix is first projected onto SO(3)
    by SVD as OpenCV does."""
    u, _, vt = np.linalg.svd(np.asarray(R, np.float64).reshape(3, 3))
    R = u @ vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r * r).sum() * 0.25)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        # theta ~ pi: (R + I) / 2 = a a^T; take the column of the largest diagonal entry
        Bm = (R + np.eye(3)) * 0.5
        k = int(np.argmax(np.diag(Bm)))
        return theta * Bm[:, k] / np.sqrt(Bm[k, k])
    return r * (0.5 / s) * theta


class TrackDataset:
    """``Tracker.dataset``.  Built by Tracker with its engine and normalisers."""

    def __init__(self, engine, images_mean, images_std, dataset_info, trans_normalizer=0.03,
                 rot_normalizer=5 * np.pi / 180, mode="eval"):
        self.engine = engine
        self.mean = np.asarray(images_mean, np.float64)
        self.std = np.asarray(images_std, np.float64)
        self.dataset_info = dataset_info
        self.trans_normalizer = trans_normalizer
        self.rot_normalizer = rot_normalizer
        self.mode = mode
        self._dev = "cuda:%d" % engine.device

    def __len__(self):
        return 0     # the eval-mode dataset of predict.py:191 is built on root '' and has no files either

    def processData(self, rgbA, depthA, A_in_cam, rgbB, depthB, B_in_cam, maskB=None, original_size=None):
        """datasets.py:115-156.  rgbA / rgbB: [176,176,3] uint8, depthA / depthB: [176,176] uint16 mm (B already
        cropped by crop_bbox).  Returns (sample, [trans_label, rot_label], rgbA_viz, rgbB_viz, maskA, maskB) with
        sample = [dataA, dataB], float32 CPU tensors [4,176,176] (R,G,B,D normalised), as ToTensor returns them."""
        rgbA = np.ascontiguousarray(rgbA); rgbB = np.ascontiguousarray(rgbB)
        res = rgbA.shape[0]
        assert rgbA.shape == (res, res, 3) == rgbB.shape and res == 176, "processData: expects the 176x176 crops"
        depthA = np.asarray(depthA); depthB = np.asarray(depthB)
        maskA = (depthA > 100).astype(np.uint8)
        if maskB is None:
            maskB = (depthB > 100).astype(np.uint8)
        A_in_cam = np.asarray(A_in_cam, np.float64)
        dev = self._dev
        up = lambda a: torch.from_numpy(a).to(dev)   # noqa: E731
        d16 = lambda d: np.ascontiguousarray(d, dtype=np.uint16).view(np.int16)   # noqa: E731
        z_mm = float(A_in_cam[2, 3]) * 1000
        crops = [dict(rgb=up(rgbA.astype(np.uint8)), depth=up(d16(depthA)), window=(0, 0, res, res), z_offset_mm=z_mm, stats=0),
                 dict(rgb=up(rgbB.astype(np.uint8)), depth=up(d16(depthB)), window=(0, 0, res, res), z_offset_mm=z_mm, stats=1)]
        out = torch.empty((2, res, res, 4), dtype=torch.float32, device=dev)
        self.engine.preprocess(crops, out)
        chw = out.permute(0, 3, 1, 2).contiguous().cpu()
        sample = [chw[0], chw[1]]
        # label math of :138-150 (what a caller with ground truth would train against; unused at inference)
        B_in_cam = np.asarray(B_in_cam, np.float64)
        trans_label = (B_in_cam[:3, 3] - A_in_cam[:3, 3]) / self.trans_normalizer
        A2B = B_in_cam[:3, :3].dot(A_in_cam[:3, :3].T)
        A2B = A2B / np.linalg.norm(A2B, axis=0, keepdims=True)            # Utils.py:363-367
        rot_label = rotation_matrix_to_rotvec(A2B) / self.rot_normalizer
        return sample, [trans_label, rot_label], rgbA.astype(np.uint8), rgbB.astype(np.uint8), maskA, maskB

    def processPredict(self, A_in_cam, predB, original_size=None):
        """datasets.py:159-175: predB = (trans, rot) float32 [3] each -> 4x4 float64 object-in-camera pose."""
        return pose_update_host(A_in_cam, np.asarray(predB[0]), np.asarray(predB[1]), self.trans_normalizer,
                                self.rot_normalizer)
