"""Import the *unmodified* reference modules from /root/reference in this container
(TEST INFRASTRUCTURE ONLY; never runs on the GPU box, where /root/reference is absent).

The reference imports packages that are not installed offline (cv2, torchvision, open3d,
transformations, trimesh).  On the hot path it only *uses* two OpenCV calls:
``cv2.resize(..., INTER_NEAREST)`` (Utils.py:343-344) and ``cv2.Rodrigues``
(datasets.py:148,173).  We install in-memory stub modules: empty ones for the unused
imports, and a minimal ``cv2`` whose two functions restate OpenCV's published algorithm
(-> parity unpinned for exactly those two rules; everything else the reference computes
itself).  ``np.float`` is re-aliased (Utils.py:307 predates NumPy 1.24).
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("SE3TN_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "se3_tracknet.py"))


def _cv2_shim():
    from . import se3_oracle as O
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST = 0
    cv2.IMREAD_UNCHANGED = -1

    def resize(img, dsize, interpolation=0):
        assert interpolation == cv2.INTER_NEAREST, "shim implements INTER_NEAREST only"
        return np.ascontiguousarray(O.resize_nearest(np.asarray(img), dsize))

    def Rodrigues(src):
        src = np.asarray(src)
        if src.size == 3:
            return O.rodrigues(src.reshape(3)), None
        # matrix -> vector: only used for the (unused at inference) label math,
        # datasets.py:148
        from scipy.spatial.transform import Rotation
        R = src.reshape(3, 3).astype(np.float64)
        u, _, vt = np.linalg.svd(R)
        return Rotation.from_matrix(u @ vt).as_rotvec().reshape(3, 1), None

    cv2.resize = resize
    cv2.Rodrigues = Rodrigues
    return cv2


def install():
    """Install stubs and put the reference root first on sys.path (its ``datasets.py``
    must shadow the HuggingFace ``datasets`` package)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only tree
    if not hasattr(np, "float"):
        np.float = float
    for name in ("open3d", "transformations", "trimesh", "torchvision", "torchvision.models"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    if "cv2" not in sys.modules or not hasattr(sys.modules["cv2"], "_se3tn_shim"):
        shim = _cv2_shim()
        shim._se3tn_shim = True
        sys.modules["cv2"] = shim
    if sys.path[0] != REFERENCE_ROOT:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in ("datasets",):
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__file__", "").startswith(REFERENCE_ROOT):
            del sys.modules[name]


def load():
    """Returns a namespace with the reference modules (se3_tracknet, Utils,
    data_augmentation, datasets)."""
    install()
    ns = types.SimpleNamespace()
    for name in ("network_modules", "se3_tracknet", "Utils", "data_augmentation", "datasets"):
        setattr(ns, name, importlib.import_module(name))
    return ns
