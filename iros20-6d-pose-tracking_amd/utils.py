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


def load_ply_mesh(path):
    """Vertices, faces, vertex colours and normals of a .ply (ascii or binary_little_endian), the
    fields VispyRenderer reads (vispy_renderer.py:113-127)."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt, elems, cur = None, [], None
        while True:
            line = f.readline().strip().decode()
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = dict(name=tok[1], count=int(tok[2]), props=[])
                elems.append(cur)
            elif tok[0] == "property":
                cur["props"].append(tok[1:])
            elif tok[0] == "end_header":
                break
        code = {"float": "f", "float32": "f", "double": "d", "float64": "d", "uchar": "B", "uint8": "B", "char": "b",
                "int8": "b", "int": "i", "int32": "i", "uint": "I", "uint32": "I", "short": "h", "int16": "h",
                "ushort": "H", "uint16": "H"}
        data = {}
        for el in elems:
            names = [p_[-1] for p_ in el["props"]]
            if el["name"] == "vertex":
                if fmt == "ascii":
                    arr = np.loadtxt(f, max_rows=el["count"], ndmin=2)
                else:
                    st = struct.Struct("<" + "".join(code[p_[0]] for p_ in el["props"]))
                    raw = f.read(st.size * el["count"])
                    arr = np.array([st.unpack_from(raw, i * st.size) for i in range(el["count"])], np.float64)
                data["vertex"] = {n: arr[:, i] for i, n in enumerate(names)}
            elif el["name"] == "face":
                faces = []
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        t = f.readline().split()
                        faces.append([int(x) for x in t[1:1 + int(t[0])]])
                else:
                    cnt_t, idx_t = code[el["props"][0][1]], code[el["props"][0][2]]
                    for _ in range(el["count"]):
                        (k,) = struct.unpack("<" + cnt_t, f.read(struct.calcsize(cnt_t)))
                        faces.append(list(struct.unpack("<" + idx_t * k, f.read(struct.calcsize(idx_t) * k))))
                data["face"] = faces
            else:  # skip other elements (ascii only)
                for _ in range(el["count"]):
                    f.readline()
    v = data["vertex"]
    out = dict(vertices=np.stack([v["x"], v["y"], v["z"]], 1))
    tri = [t for t in data.get("face", []) if len(t) == 3]
    out["faces"] = np.asarray(tri, np.int32).reshape(-1, 3)
    if "red" in v:
        out["colors"] = np.stack([v["red"], v["green"], v["blue"]], 1)
    else:
        out["colors"] = np.full((len(out["vertices"]), 3), 255.0)
    out["normals"] = np.stack([v["nx"], v["ny"], v["nz"]], 1) if "nx" in v else None
    if out["normals"] is not None and not np.all(np.linalg.norm(out["normals"], axis=1) > 0):
        out["normals"] = None
    return out


def load_obj_mesh(path):
    """Wavefront .obj as the reference's pyrender route reads it through trimesh (offscreen_renderer.py:60-63):
    vertices, triangles (polygons fan-triangulated), texture coordinates, the material's diffuse texture (map_Kd)
    and Kd.  Vertices are split per distinct (position, texcoord) pair so that every vertex carries one uv.
    Returns dict(vertices [V,3], faces [F,3] int32, uv [V,2] | None, colors [V,3] 0..255, texture uint8 [h,w,3] | None,
    kd [3], normals None)."""
    import os
    pos, tex, col, corners, faces = [], [], [], {}, []
    mtllib = None
    out_v, out_uv, out_c = [], [], []

    def corner(tok):
        parts = tok.split("/")
        vi = int(parts[0]); vi = vi - 1 if vi > 0 else len(pos) + vi
        ti = -1
        if len(parts) > 1 and parts[1]:
            ti = int(parts[1]); ti = ti - 1 if ti > 0 else len(tex) + ti
        key = (vi, ti)
        if key not in corners:
            corners[key] = len(out_v)
            out_v.append(pos[vi]); out_c.append(col[vi])
            out_uv.append(tex[ti] if ti >= 0 else (0.0, 0.0))
        return corners[key]

    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                pos.append(tuple(float(x) for x in t[1:4]))
                col.append(tuple(float(x) * 255.0 for x in t[4:7]) if len(t) >= 7 else (255.0, 255.0, 255.0))
            elif t[0] == "vt":
                tex.append((float(t[1]), float(t[2]) if len(t) > 2 else 0.0))
            elif t[0] == "f":
                idx = [corner(tok) for tok in t[1:]]
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
            elif t[0] == "mtllib":
                mtllib = line.split(None, 1)[1].strip()
    out = dict(vertices=np.asarray(out_v, np.float64).reshape(-1, 3), faces=np.asarray(faces, np.int32).reshape(-1, 3),
               uv=np.asarray(out_uv, np.float64).reshape(-1, 2) if tex else None,
               colors=np.asarray(out_c, np.float64).reshape(-1, 3), texture=None, kd=np.ones(3), normals=None)
    if mtllib:
        mpath = os.path.join(os.path.dirname(path), mtllib)
        if os.path.isfile(mpath):
            for line in open(mpath):
                t = line.split()
                if not t:
                    continue
                if t[0] == "Kd" and len(t) >= 4:
                    out["kd"] = np.array([float(x) for x in t[1:4]])
                elif t[0] == "map_Kd":
                    tpath = os.path.join(os.path.dirname(mpath), line.split(None, 1)[1].strip())
                    if os.path.isfile(tpath):
                        from PIL import Image
                        out["texture"] = np.ascontiguousarray(np.array(Image.open(tpath).convert("RGB")), dtype=np.uint8)
    if out["texture"] is not None and np.all(out["kd"] == 0):
        out["kd"] = np.ones(3)   # exporters write Kd 0 0 0 next to a map_Kd; trimesh then takes the texture as the colour
    return out


def vertex_normals(vertices, faces):
    """Area-weighted vertex normals for meshes that store none."""
    v = np.asarray(vertices, np.float64)
    n = np.zeros_like(v)
    fn = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    for k in range(3):
        np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1)
    n[ln > 0] /= ln[ln > 0, None]
    n[ln == 0] = (0, 0, 1)
    return n


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
