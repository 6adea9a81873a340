"""Minimal PLY reading + open3d-style voxel down-sampling for the stand-ins that let the UNMODIFIED reference classes run
(oracle/swiftshader_gl.py: `plyfile.PlyData.read`; oracle/make_predict_golden.py: `open3d` / `trimesh`).

TEST INFRASTRUCTURE ONLY, and deliberately independent of the product package (the oracle of a product must
not be built from the product's own loaders): plain numpy, written for the files the fixtures write (ascii or
binary_little_endian, scalar properties + one list property per face)."""
import numpy as np

_T = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
      "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


def read_ply(path):
    """{element: {property: ndarray}}; list properties become an int64 [n, k] array (k constant) or a list of arrays."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt, elems = None, []
        while True:
            t = f.readline().decode().split()
            if not t or t[0] == "comment":
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                elems.append((t[1], int(t[2]), []))
            elif t[0] == "property":
                elems[-1][2].append(tuple(t[1:]))
            elif t[0] == "end_header":
                break
        out = {}
        for name, n, props in elems:
            cols = {p[-1]: [] for p in props}
            for _ in range(n):
                if fmt == "ascii":
                    tok = f.readline().split()
                    k = 0
                    for p in props:
                        if p[0] == "list":
                            cnt = int(tok[k]); k += 1
                            cols[p[-1]].append(np.array(tok[k:k + cnt], dtype=np.int64)); k += cnt
                        else:
                            cols[p[-1]].append(float(tok[k])); k += 1
                else:
                    assert fmt == "binary_little_endian", fmt
                    for p in props:
                        if p[0] == "list":
                            ct, it = np.dtype("<" + _T[p[1]]), np.dtype("<" + _T[p[2]])
                            cnt = int(np.frombuffer(f.read(ct.itemsize), ct)[0])
                            cols[p[-1]].append(np.frombuffer(f.read(it.itemsize * cnt), it).astype(np.int64))
                        else:
                            dt = np.dtype("<" + _T[p[0]])
                            cols[p[-1]].append(np.frombuffer(f.read(dt.itemsize), dt)[0])
            res = {}
            for p in props:
                v = cols[p[-1]]
                if p[0] == "list":
                    res[p[-1]] = np.stack(v) if v and all(len(x) == len(v[0]) for x in v) else v
                else:
                    res[p[-1]] = np.asarray(v, dtype=_T[p[0]])
            out[name] = res
    return out


def ply_vertices(path):
    v = read_ply(path)["vertex"]
    return np.stack([np.asarray(v[k], np.float64) for k in ("x", "y", "z")], 1)


def voxel_down_sample(points, voxel_size):
    """open3d.geometry.PointCloud.voxel_down_sample as open3d documents it: voxel grid anchored at min_bound - voxel_size / 2,
    one output point per occupied voxel = the mean of its points."""
    pts = np.asarray(points, np.float64)
    key = np.floor((pts - (pts.min(0) - 0.5 * voxel_size)) / voxel_size).astype(np.int64)
    order = np.lexsort((key[:, 2], key[:, 1], key[:, 0]))
    ks, ps = key[order], pts[order]
    first = np.r_[True, (ks[1:] != ks[:-1]).any(1)]
    idx = np.cumsum(first) - 1
    out = np.zeros((idx[-1] + 1, 3))
    np.add.at(out, idx, ps)
    return out / np.bincount(idx)[:, None]
