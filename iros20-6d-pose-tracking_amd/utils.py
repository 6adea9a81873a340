"""Host-side helpers of the hot path that the reference keeps in Utils.py, plus dependency-free
stand-ins for the trimesh / open3d calls of Tracker.__init__ (predict.py:131-142)."""
import struct

import numpy as np

from .engine import compute_bbox as _compute_bbox_c


def compute_bbox(pose, K, scale_size=230, scale=(1000, 1000, 1000)):
    """Utils.py:302-316.  The path only ever calls it with scale (1000,1000,1000)
    (predict.py:232) or (1000,-1000,1000) for the GL renderer window (predict.py:201)."""
    if tuple(scale) == (1000, 1000, 1000):
        return _compute_bbox_c(pose, K, scale_size)
    # generic float64 numpy path for the renderer's flipped-y window
    obj = [pose[i, 3] * scale[i] for i in range(3)]
    off = scale_size / 2
    pts = np.array([[obj[0] - off, obj[1] - off, obj[2]], [obj[0] - off, obj[1] + off, obj[2]],
                    [obj[0] + off, obj[1] - off, obj[2]], [obj[0] + off, obj[1] + off, obj[2]]], np.float64)
    vus = np.zeros((4, 2))
    vus[:, 1] = pts[:, 0] * K[0, 0] / pts[:, 2] + K[0, 2]
    vus[:, 0] = pts[:, 1] * K[1, 1] / pts[:, 2] + K[1, 2]
    return np.round(vus).astype(np.int32)


def crop_window(bbox):
    """left, top, right, bottom as crop_bbox derives them (Utils.py:321-324)."""
    return (int(np.min(bbox[:, 1])), int(np.min(bbox[:, 0])), int(np.max(bbox[:, 1])), int(np.max(bbox[:, 0])))


class PointCloud:
    """Minimal stand-in for the open3d cloud kept in Tracker.object_cloud (callers read
    np.asarray(tracker.object_cloud.points), predict.py:424,549)."""
    def __init__(self, points):
        self.points = np.asarray(points, np.float64)


def load_model_points(path):
    """Vertices of a .ply (ascii / binary_little_endian) or .obj model."""
    if path.endswith(".obj"):
        pts = [list(map(float, l.split()[1:4])) for l in open(path) if l.startswith("v ")]
        return np.asarray(pts, np.float64)
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt, nv, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline().strip().decode()
            if line.startswith("format"):
                fmt = line.split()[1]
            elif line.startswith("element"):
                in_vertex = line.split()[1] == "vertex"
                if in_vertex:
                    nv = int(line.split()[2])
            elif line.startswith("property") and in_vertex:
                props.append((line.split()[1], line.split()[2]))
            elif line == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=nv, ndmin=2)
            idx = [i for i, p in enumerate(props) if p[1] in ("x", "y", "z")]
            return data[:, idx].astype(np.float64)
        code = {"float": "f", "float32": "f", "double": "d", "float64": "d", "uchar": "B", "uint8": "B",
                "char": "b", "int": "i", "int32": "i", "uint": "I", "short": "h", "ushort": "H"}
        st = struct.Struct("<" + "".join(code[p[0]] for p in props))
        raw = f.read(st.size * nv)
        rows = np.array([st.unpack_from(raw, i * st.size) for i in range(nv)], np.float64)
        idx = [i for i, p in enumerate(props) if p[1] in ("x", "y", "z")]
        return rows[:, idx]


def voxel_down_sample(points, voxel_size=0.005):
    """open3d PointCloud.voxel_down_sample restated: grid origin = min_bound - voxel/2, one
    averaged point per occupied voxel (predict.py:133)."""
    pts = np.asarray(points, np.float64)
    origin = pts.min(0) - voxel_size * 0.5
    idx = np.floor((pts - origin) / voxel_size).astype(np.int64)
    _, inv = np.unique(idx, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    out = np.zeros((inv.max() + 1, 3))
    np.add.at(out, inv, pts)
    return out / np.bincount(inv)[:, None]


def compute_obj_max_width(points):
    """Utils.py:101-105,450-451: convex-hull diameter in millimetres."""
    from scipy.spatial import ConvexHull, distance_matrix
    hull = ConvexHull(points)
    hp = points[hull.vertices]
    return float(np.max(distance_matrix(hp, hp))) * 1000
