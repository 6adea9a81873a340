"""Importable alias of the package directory ``iros20-6d-pose-tracking_amd`` (its name is not a
valid Python identifier):   import se3tracknet_amd as se3"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("iros20-6d-pose-tracking_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
engine = importlib.import_module("iros20-6d-pose-tracking_amd.engine")
utils = importlib.import_module("iros20-6d-pose-tracking_amd.utils")
_lib = importlib.import_module("iros20-6d-pose-tracking_amd._lib")
package = _pkg
