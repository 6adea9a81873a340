"""Drop-in for the reference's ``se3_tracknet.Se3TrackNet`` at inference time
(se3_tracknet.py:52-112): same constructor, ``load_state_dict`` surface, ``.cuda()``,
``.eval()`` and ``model(A, B) -> {'feature','trans','rot'}`` on float32 CUDA tensors
[N,4,176,176] (predict.py:155-158, 270-276) -- executed by the gfx950 HIP library."""
import torch

from .engine import Engine, NCHW, NHWC


class Se3TrackNet:
    def __init__(self, image_size=176, max_batch=64):
        if image_size != 176:
            raise ValueError("the HIP path is specialised for dataset_info['resolution'] == 176")
        self.rot_dim = 3
        self.max_batch = max_batch
        self._sd = None
        self.engine = None

    @classmethod
    def from_engine(cls, engine):
        """The callable ``Tracker.model`` (predict.py:270-271): shares the Tracker's engine and weights."""
        m = cls(176, engine.max_batch)
        m.engine = engine
        return m

    def load_state_dict(self, state_dict, strict=True):
        self._sd = state_dict
        if self.engine is not None:
            self.engine.load_state_dict(state_dict)
        return self

    def cuda(self, device=None):
        dev = torch.cuda.current_device() if device is None else int(device)
        self.engine = Engine(dev, self.max_batch)
        if self._sd is not None:
            self.engine.load_state_dict(self._sd)
        return self

    def eval(self):
        return self

    def __call__(self, A, B, return_feature=True):
        return self.forward(A, B, return_feature)

    def forward(self, A, B, return_feature=True):
        if self.engine is None or not self.engine.has_weights:
            raise RuntimeError("Se3TrackNet: call load_state_dict(...) and .cuda() first")
        assert A.is_cuda and B.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32
        assert A.shape == B.shape and tuple(A.shape[1:]) == (4, 176, 176)
        n = A.shape[0]
        A = A.contiguous(); B = B.contiguous()
        trans = torch.empty((n, 3), dtype=torch.float32, device=A.device)
        rot = torch.empty((n, 3), dtype=torch.float32, device=A.device)
        self.engine.infer(A, B, n, NCHW, trans, rot)
        out = {"trans": trans, "rot": rot}
        if return_feature:
            out["feature"] = self.engine.feature(n)
        return out
