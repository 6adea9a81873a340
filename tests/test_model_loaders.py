"""CPU: the model loaders behind Tracker.__init__ (predict.py:131-142, vispy_renderer.py:113-127, offscreen_renderer.py:60-63)
are vectorised -- a mesh of YCB size (>= 250 k vertices, 500 k faces, binary PLY / textured OBJ) loads in seconds, not tens of
seconds -- and agree with a plain per-record parse."""
import struct
import time

import numpy as np
import pytest


@pytest.fixture(scope="module")
def U():
    import se3tracknet_amd
    return se3tracknet_amd.utils


def _big_mesh(nv=262_144, seed=0):
    rng = np.random.default_rng(seed)
    v = rng.uniform(-0.1, 0.1, (nv, 3)).astype(np.float32)
    n = rng.normal(size=(nv, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    c = rng.integers(0, 256, (nv, 3), dtype=np.uint8)
    f = rng.integers(0, nv, (2 * nv, 3), dtype=np.int32)
    return v, n, c, f


def _write_binary_ply(path, v, n, c, f, big_endian=False, ragged=False):
    bo = ">" if big_endian else "<"
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_%s_endian 1.0\ncomment made by tests\nelement vertex %d\nproperty float x\nproperty float y\n"
                  "property float z\nproperty float nx\nproperty float ny\nproperty float nz\nproperty uchar red\nproperty uchar green\n"
                  "property uchar blue\nproperty uchar alpha\nelement face %d\nproperty list uchar int vertex_indices\nend_header\n"
                  % ("big" if big_endian else "little", len(v), len(f) + (1 if ragged else 0))).encode())
        rec = np.zeros(len(v), dtype=[("p", bo + "f4", 3), ("n", bo + "f4", 3), ("c", "u1", 4)])
        rec["p"], rec["n"], rec["c"][:, :3], rec["c"][:, 3] = v, n, c, 255
        fh.write(rec.tobytes())
        fr = np.zeros(len(f), dtype=[("k", "u1"), ("i", bo + "i4", 3)])
        fr["k"], fr["i"] = 3, f
        fh.write(fr.tobytes())
        if ragged:
            fh.write(struct.pack(bo + "B4i", 4, 0, 1, 2, 3))      # one quad at the end: the element is no longer uniform


def _slow_ply_vertices(path, count):
    """per-record struct.unpack of the first `count` vertices (what the loaders did before)."""
    with open(path, "rb") as fh:
        while fh.readline().strip() != b"end_header":
            pass
        st = struct.Struct("<6f4B")
        raw = fh.read(st.size * count)
    return np.array([st.unpack_from(raw, i * st.size) for i in range(count)], np.float64)


def test_binary_ply_of_ycb_size_loads_fast_and_exact(U, tmp_path):
    v, n, c, f = _big_mesh()
    path = str(tmp_path / "big.ply")
    _write_binary_ply(path, v, n, c, f)
    t0 = time.perf_counter()
    m = U.load_ply_mesh(path)
    pts = U.load_model_points(path)
    dt = time.perf_counter() - t0
    assert dt < 5.0, "vectorised loaders took %.1f s for 262 k vertices / 524 k faces" % dt
    assert m["vertices"].shape == (len(v), 3) and m["faces"].shape == (len(f), 3) and m["faces"].dtype == np.int32
    assert np.array_equal(m["vertices"], v.astype(np.float64)) and np.array_equal(pts, v.astype(np.float64))
    assert np.array_equal(m["faces"], f) and np.array_equal(m["colors"], c.astype(np.float64))
    assert np.allclose(m["normals"], n)
    slow = _slow_ply_vertices(path, 2000)
    assert np.array_equal(slow[:, :3], m["vertices"][:2000]) and np.array_equal(slow[:, 6:9], m["colors"][:2000])
    # object_width of Tracker.__init__ (predict.py:136-142) from the down-sampled cloud: bounded time, same value as from a
    # per-record parse of the same file
    t0 = time.perf_counter()
    ds = U.voxel_down_sample(pts, 0.005)
    w = U.compute_obj_max_width(ds)
    assert time.perf_counter() - t0 < 20.0
    st = struct.Struct("<6f4B")
    with open(path, "rb") as fh:
        while fh.readline().strip() != b"end_header":
            pass
        raw = fh.read(st.size * len(v))
    slow_pts = np.frombuffer(raw, dtype=np.dtype([("p", "<f4", 3), ("r", "u1", 16)]))["p"].astype(np.float64)
    assert abs(U.compute_obj_max_width(U.voxel_down_sample(slow_pts, 0.005)) - w) < 1e-9


def test_big_endian_and_ragged_faces(U, tmp_path):
    v, n, c, f = _big_mesh(3000, 1)
    pb = str(tmp_path / "be.ply")
    _write_binary_ply(pb, v, n, c, f, big_endian=True)
    m = U.load_ply_mesh(pb)
    assert np.array_equal(m["vertices"], v.astype(np.float64)) and np.array_equal(m["faces"], f)
    pr = str(tmp_path / "rag.ply")
    _write_binary_ply(pr, v, n, c, f, ragged=True)
    m = U.load_ply_mesh(pr)                              # the quad is dropped (the rasteriser takes triangles)
    assert np.array_equal(m["faces"], f) and np.array_equal(m["vertices"], v.astype(np.float64))


def test_ascii_ply_with_faces_and_extra_element(U, tmp_path):
    v, n, c, f = _big_mesh(200, 2)
    p = tmp_path / "a.ply"
    with open(p, "w") as fh:
        fh.write("ply\nformat ascii 1.0\nelement vertex 200\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n"
                 "property uchar green\nproperty uchar blue\nelement face 400\nproperty list uchar int vertex_indices\n"
                 "element edge 2\nproperty int a\nproperty int b\nend_header\n")
        for i in range(200):
            fh.write("%.9g %.9g %.9g %d %d %d\n" % (*v[i], *c[i]))
        for t in f:
            fh.write("3 %d %d %d\n" % tuple(t))
        fh.write("0 1\n1 2\n")
    m = U.load_ply_mesh(str(p))
    assert np.allclose(m["vertices"], v, atol=1e-7) and np.array_equal(m["faces"], f) and m["normals"] is None
    assert np.array_equal(m["colors"], c.astype(np.float64))


def test_obj_fast_parser_equals_general_parser(U, tmp_path):
    rng = np.random.default_rng(4)
    nv, nt = 60_000, 70_000
    v = rng.uniform(-0.1, 0.1, (nv, 3))
    vt = rng.uniform(0, 1, (nt, 2))
    fv = rng.integers(1, nv + 1, (120_000, 3))
    ft = rng.integers(1, nt + 1, (120_000, 3))
    for fmt in ("v/vt", "v", "v/vt/vn", "v//vn"):
        p = tmp_path / ("m_%s.obj" % fmt.replace("/", "_"))
        with open(p, "w") as fh:
            fh.write("# test\no thing\n")
            fh.write("".join("v %.9f %.9f %.9f\n" % tuple(x) for x in v))
            fh.write("".join("vt %.9f %.9f\n" % tuple(x) for x in vt))
            fh.write("vn 0 0 1\ns off\n")
            if fmt == "v/vt":
                fh.write("".join("f %d/%d %d/%d %d/%d\n" % (a[0], b[0], a[1], b[1], a[2], b[2]) for a, b in zip(fv, ft)))
            elif fmt == "v":
                fh.write("".join("f %d %d %d\n" % tuple(a) for a in fv))
            elif fmt == "v/vt/vn":
                fh.write("".join("f %d/%d/1 %d/%d/1 %d/%d/1\n" % (a[0], b[0], a[1], b[1], a[2], b[2]) for a, b in zip(fv, ft)))
            else:
                fh.write("".join("f %d//1 %d//1 %d//1\n" % tuple(a) for a in fv))
        lines = open(p).read().splitlines()
        assert U._obj_geometry_fast(lines) is not None, fmt
        t0 = time.perf_counter()
        fast = U.load_obj_mesh(str(p))
        t_fast = time.perf_counter() - t0
        slow = U._load_obj_geometry_general(lines)
        for k in ("vertices", "faces", "colors"):
            assert np.array_equal(fast[k], slow[k]), (fmt, k)
        assert (fast["uv"] is None) == (slow["uv"] is None)
        if fast["uv"] is not None:
            assert np.array_equal(fast["uv"], slow["uv"]), fmt
        assert t_fast < 5.0
    # files the fast parser must hand over: quads, negative indices, vertex colours with w
    for txt in ("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n", "v 0 0 0\nv 1 0 0\nv 1 1 0\nf -3 -2 -1\n",
                "v 0 0 0\nv 1 0 0\nv 1 1 0\nf 1/ 2/ 3/\n"):
        assert U._obj_geometry_fast(txt.splitlines()) is None
    # mixed corner formats / separators whose FLATTENED token counts happen to match one format: the fast parser either
    # declines or agrees with the general parser -- never a silently different mesh
    tri = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 2 2 0\nv 3 3 0\nvt 0 0\nvt 1 0\nvt 1 1\n"
    for body in ("f 1/1 2/2 3/3\nf 1 2 3 4 5 6\n",             # 2 x 6 ints either way
                 "f 1/1/1 2/2/2 3/3/3\nf 1//1 2//2 3//3\n",     # same '/' count per line, different meaning
                 "f 1/1 2/2 3/3\nf 1 2 3\nf 4 5 6\n",
                 "f\t1/1\t2/2\t3/3\n  f 2/2 3/3 4/1\n",          # tab-separated and indented records
                 "f 1 2 3\n\tv 9 9 9\nf 2 3 7\n"):              # an indented vertex record between the faces
        lines = (tri + body).splitlines()
        fast = U._obj_geometry_fast(lines)
        if fast is None:
            continue
        (tmp_path / "mix.obj").write_text(tri + body)
        got, want = U.load_obj_mesh(str(tmp_path / "mix.obj")), U._load_obj_geometry_general(lines)
        for k in ("vertices", "faces", "colors"):
            assert np.array_equal(got[k], want[k]), (body, k)
    # corner formats mixed INSIDE one face line: 3 tokens, 3 slashes, 6 integers like "1/3 2/2 1/4" 
    mixed = (tri + "f 1/3 2/2/1 4\n").splitlines()
    assert U._obj_geometry_fast(mixed) is None
    assert U._load_obj_geometry_general(mixed)["faces"].shape == (1, 3)
    assert U._obj_geometry_fast((tri + "f 1/1 2/2 3/3\nf 1 2 3 4 5 6\n").splitlines()) is None
    assert U._obj_geometry_fast((tri + "f 1/1/1 2/2/2 3/3/3\nf 1//1 2//2 3//3\n").splitlines()) is None
    (tmp_path / "vc.obj").write_text("v 0 0 0 1 0 0\nv 1 0 0 0 1 0\nv 1 1 0 0 0 1\nf 1 2 3\n")
    m = U.load_obj_mesh(str(tmp_path / "vc.obj"))
    assert m["colors"].tolist() == [[255.0, 0, 0], [0, 255.0, 0], [0, 0, 255.0]]
