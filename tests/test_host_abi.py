"""CPU: the C-ABI library loads without a GPU, exports every symbol include/se3tracknet.h declares,
and its host-side (float64 / packing / error-path) logic matches the oracle bit for bit."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import fixtures as Fx
from oracle import se3_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def se3():
    import se3tracknet_amd
    return se3tracknet_amd


def test_library_exports_every_declared_symbol(se3):
    hdr = open(os.path.join(ROOT, "include", "se3tracknet.h")).read()
    declared = sorted(set(re.findall(r"\b(se3tn_[a-z0-9_]+)\s*\(", hdr)))
    lib = C.CDLL(se3._lib.LIB_PATH)
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert sorted(se3._lib.exported_symbols()) == declared  # ctypes table == header
    assert b"gfx950" in se3._lib.load().se3tn_version()


def test_compute_bbox_bit_exact_vs_oracle(se3):
    rng = np.random.default_rng(0)
    for i in range(300):
        P = Fx.pose(i, (rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), rng.uniform(0.25, 2.0)))
        w = float(rng.uniform(80, 400))
        got = se3.compute_bbox(P, Fx.K_YCB, w)
        want = O.compute_bbox(P, Fx.K_YCB, w, (1000, 1000, 1000))
        assert got.dtype == np.int32 and (got == want).all(), (i, got, want)
    # half-to-even rounding, as np.round: K chosen so that u lands exactly on .5
    P = np.eye(4); P[2, 3] = 1.0
    K = np.array([[1.0, 0, 0.5], [0, 1.0, 1.5], [0, 0, 1.0]])
    assert (se3.compute_bbox(P, K, 0.0) == O.compute_bbox(P, K, 0.0, (1000, 1000, 1000))).all()
    assert se3.compute_bbox(P, K, 0.0)[0].tolist() == [2, 0]  # 1.5 -> 2, 0.5 -> 0


def test_renderer_window_variant_matches_oracle(se3):
    P = Fx.pose(5)
    got = se3.compute_bbox(P, Fx.K_YCB, 250.0, scale=(1000, -1000, 1000))
    assert (got == O.compute_bbox(P, Fx.K_YCB, 250.0, (1000, -1000, 1000))).all()


def test_pose_update_host_bit_exact_vs_reference_golden(se3, golden_dir):
    g = np.load(os.path.join(golden_dir, "pose_update.npz"))
    for i in range(len(g["A"])):
        B = se3.pose_update_host(g["A"][i], g["trans"][i], g["rot"][i], float(g["norm"][i][0]), float(g["norm"][i][1]))
        assert np.abs(B - g["B"][i]).max() <= 2.3e-16, (i, np.abs(B - g["B"][i]).max())  # libm sin/cos ulp
        assert (B[3] == [0, 0, 0, 1]).all()
    assert (se3.pose_update_host(g["A"][0], g["trans"][0], np.zeros(3, np.float32), 0.03, 0.1)[:3, :3] == g["A"][0][:3, :3]).all()


def _numpy_pack(sd):
    """Independent restatement of the blob layout (float32 panels) and of the f16x3 split layout documented in
    csrc/se3tn_internal.h (BlobLayout / SplitLayout).  Returns (blob, split) as float32 word arrays."""
    def fold(conv, bn):
        w = sd[conv + ".weight"].double().numpy(); b = sd[conv + ".bias"].double().numpy()
        s = sd[bn + ".weight"].double().numpy() / np.sqrt(sd[bn + ".running_var"].double().numpy() + 1e-5)
        return (w * s[:, None, None, None]).astype(np.float32), ((b - sd[bn + ".running_mean"].double().numpy()) * s + sd[bn + ".bias"].double().numpy()).astype(np.float32)

    def pack3(w):  # OIHW -> [chunk][tap][cout][32]
        co, ci = w.shape[:2]
        return w.reshape(co, ci // 32, 32, 9).transpose(1, 3, 0, 2).reshape(-1)

    parts = [np.zeros(64, np.float32)]
    stems = [fold(n + ".0", n + ".1") for n in ("convA1", "convB1")]
    pairs = [((r, 2 * sp), (r, 2 * sp + 1)) for r in range(7) for sp in range(3)]
    pairs += [((2 * j, 6), (2 * j + 1, 6)) for j in range(3)] + [((6, 6), None)]
    stem_plain = []
    for w, _ in stems:  # [o][pair*8 + half*4 + c], row padded to 204
        t = np.zeros((64, 204), np.float32)
        for pi, pr in enumerate(pairs):
            for h, tap in enumerate(pr):
                if tap is not None:
                    t[:, pi * 8 + h * 4: pi * 8 + h * 4 + 4] = w[:, :, tap[0], tap[1]]
        parts.append(t.reshape(-1))
        stem_plain.append(t)
    parts += [b for _, b in stems]
    split = []
    inv = []
    for t in stem_plain:  # f16x3 stem: per 16-byte entry 4 f16 hi | 4 f16 lo of w * 2^k(cout)
        mx = np.abs(t[:, :200]).max(1).astype(np.float64)
        k = np.where(mx > 0, np.floor(10.0 - np.log2(np.where(mx > 0, mx, 1.0))), 0.0)
        ws = t[:, :200] * np.exp2(k).astype(np.float32)[:, None]
        hi = ws.astype(np.float16); lo = (ws - hi.astype(np.float32)).astype(np.float16)
        e = np.concatenate([hi.reshape(64, 50, 4), lo.reshape(64, 50, 4)], axis=2).reshape(64, 400)
        row = np.zeros((64, 408), np.float16); row[:, :400] = e
        split.append(np.ascontiguousarray(row).view(np.float32).reshape(-1))
        inv.append(np.exp2(-k).astype(np.float32))
    split += inv
    groups = [[("convA2.conv1", "convA2.bn1"), ("convB2.conv1", "convB2.bn1")],
              [("convA2.conv2", "convA2.bn2"), ("convB2.conv2", "convB2.bn2")],
              [("convB3.conv1", "convB3.bn1")], [("convB3.conv2", "convB3.bn2")],
              [("convAB1.0", "convAB1.1")], [("convAB2.conv1", "convAB2.bn1")], [("convAB2.conv2", "convAB2.bn2")],
              "H1",
              [("trans_conv2.conv1", "trans_conv2.bn1"), ("rot_conv2.conv1", "rot_conv2.bn1")],
              [("trans_conv2.conv2", "trans_conv2.bn2"), ("rot_conv2.conv2", "rot_conv2.bn2")]]
    for g in groups:
        if g == "H1":  # trans|rot fused along Cout
            (wt, bt), (wr, br) = fold("trans_conv1.0", "trans_conv1.1"), fold("rot_conv1.0", "rot_conv1.1")
            parts += [pack3(np.concatenate([wt, wr], 0)), bt, br]
        else:
            f = [fold(c, b) for c, b in g]
            parts += [pack3(w) for w, _ in f] + [b for _, b in f]
    # f16x3 sections: the deep layers again as split rows [chunk][tap][cout][32 f16 hi | 32 f16 lo] of
    # w * 2^k(cout) plus the per-cout 2^-k
    def pack3_split(w):
        co, ci = w.shape[:2]
        mx = np.abs(w).reshape(co, -1).max(1).astype(np.float64)
        k = np.where(mx > 0, np.floor(10.0 - np.log2(np.where(mx > 0, mx, 1.0))), 0.0)
        ws = (w.astype(np.float32) * np.exp2(k).astype(np.float32)[:, None, None, None])
        hi = ws.astype(np.float16)
        lo = (ws - hi.astype(np.float32)).astype(np.float16)
        def rows(x):  # OIHW -> [chunk][tap][cout][32]
            return x.reshape(co, ci // 32, 32, 9).transpose(1, 3, 0, 2)
        both = np.concatenate([rows(hi), rows(lo)], axis=3)  # [..][64] halfs = 128 bytes
        return np.ascontiguousarray(both).view(np.float32).reshape(-1), np.exp2(-k).astype(np.float32)
    for g in groups:
        if g == "H1":
            (wt, _), (wr, _) = fold("trans_conv1.0", "trans_conv1.1"), fold("rot_conv1.0", "rot_conv1.1")
            ws, sc = pack3_split(np.concatenate([wt, wr], 0))
            split += [ws, sc]
        else:
            f = [pack3_split(fold(c, b)[0]) for c, b in g]
            split += [x[0] for x in f] + [x[1] for x in f]
    for h in ("trans_out", "rot_out"):
        parts.append(sd[h + ".0.weight"].numpy().reshape(-1))
    for h in ("trans_out", "rot_out"):
        parts.append(np.concatenate([sd[h + ".0.bias"].numpy(), np.zeros(1, np.float32)]))
    blob = np.concatenate(parts)
    split = np.concatenate(split)
    pad = lambda x: np.concatenate([x, np.zeros((-len(x)) % 64, np.float32)])   # noqa: E731
    return pad(blob), pad(split)


def test_weight_folding_and_packing_host_only(se3):
    eng = se3.Engine(device=-1, max_batch=1)  # host-only context: no GPU needed
    sd = O.make_state_dict(3)
    blob_t = eng.pack_state_dict(sd)
    blob = blob_t.numpy().view(np.float32)
    want, want_split = _numpy_pack(sd)
    assert blob.size == want.size == eng.packed_bytes() // 4
    assert 54.0e6 < eng.packed_bytes() < 54.3e6      # float32 panels only: what the RCCL broadcast carries (blob v7)
    hdr = blob[:4].view(np.uint32)
    assert hdr[0] == 0x53453354 and hdr[1] == 7 and hdr[2] == blob.size
    assert (blob[64:].view(np.uint32) == want[64:].view(np.uint32)).all()  # bit-exact
    # the f16x3 split panels are no longer in the blob: derived from it (on the device in the product; the library's host
    # statement of the same arithmetic here) -- bit-exact against the independent numpy restatement from the OIHW weights
    split = eng.split_weights_host(blob_t).numpy().view(np.float32)
    assert split.size == want_split.size
    assert (split.view(np.uint32) == want_split.view(np.uint32)).all()


def test_split_exponent_rule_at_powers_of_two(se3):
    """k = floor(10 - log2(max|w|)) is evaluated on the float's exponent: exact at powers of two and their neighbours."""
    eng = se3.Engine(device=-1, max_batch=1)
    sd = O.make_state_dict(5)
    w = sd["convAB2.conv1.weight"].clone()
    for i, mx in enumerate([1.0, 0.5, np.nextafter(np.float32(0.5), np.float32(1)), np.nextafter(np.float32(0.5), np.float32(0)),
                            2.0 ** -12, 3.0, 1024.0, 2047.9, 2048.0]):
        w[i] = w[i] * 1e-3
        w[i, 0, 0, 0] = float(mx)
    sd["convAB2.conv1.weight"] = w
    # fold with identity BN so that the row maxima survive exactly
    for k in ("weight", "bias", "running_mean", "running_var"):
        sd["convAB2.bn1." + k] = torch.ones(256) if k in ("weight",) else torch.zeros(256)
    sd["convAB2.bn1.running_var"] = torch.ones(256) - 1e-5
    blob_t = eng.pack_state_dict(sd)
    _, want_split = _numpy_pack(sd)
    split = eng.split_weights_host(blob_t).numpy().view(np.float32)
    assert (split.view(np.uint32) == want_split.view(np.uint32)).all()


def test_error_paths(se3):
    lib = se3._lib.load()
    eng = se3.Engine(device=-1, max_batch=1)
    t = torch.zeros(3, 3)
    sh = (C.c_int64 * 2)(3, 3)
    assert lib.se3tn_set_tensor(eng._h, b"not.a.key", C.c_void_p(t.data_ptr()), sh, 2) == -4  # SE3TN_E_KEY
    assert b"not.a.key" in lib.se3tn_last_error()
    assert lib.se3tn_set_tensor(eng._h, b"trans_out.0.weight", C.c_void_p(t.data_ptr()), sh, 2) == -3  # SE3TN_E_SHAPE
    assert lib.se3tn_pack_weights(eng._h) == -4 and b"missing" in lib.se3tn_last_error()
    with pytest.raises(se3._lib.Se3tnError):  # int64 / float64 tensors are refused, not silently cast
        eng.pack_state_dict({"trans_out.0.bias": torch.zeros(3, dtype=torch.float64)})
    h = C.c_void_p()
    assert lib.se3tn_create(0, 0, C.byref(h)) == -1  # bad max_batch
    if not torch.cuda.is_available():
        assert lib.se3tn_create(0, 1, C.byref(h)) != 0  # no device: loud failure, no fallback
        with pytest.raises(se3._lib.Se3tnError):
            se3.Engine(0, 1)
    # compute entry points refuse a host-only context
    assert lib.se3tn_infer(eng._h, C.c_void_p(8), C.c_void_p(8), 1, 0, None, None, None, None, None) == -1


def test_trunk_winograd_switch_defaults_and_argument_checks(se3):
    """se3tn_set/get_trunk_winograd on a host-only context: the defaults are the header's, the ctypes mirror carries the same
    numbers, bad arguments are refused."""
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "se3tracknet.h")).read()
    mb = int(re.search(r"#define SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_BATCH (\d+)", hdr).group(1))
    mf = int(re.search(r"#define SE3TN_TRUNK_WINOGRAD_DEFAULT_MIN_FILL (\d+)", hdr).group(1))
    assert (mb, mf) == (se3._lib.TRUNK_WINOGRAD_DEFAULT_MIN_BATCH, se3._lib.TRUNK_WINOGRAD_DEFAULT_MIN_FILL)
    eng = se3.Engine(device=-1, max_batch=1)
    assert eng.get_trunk_winograd() == (mb, mf)
    eng.set_trunk_winograd(0)
    assert eng.get_trunk_winograd() == (0, mf)
    eng.set_trunk_winograd(16, 0)
    assert eng.get_trunk_winograd() == (16, 0)
    lib = se3._lib.load()
    assert lib.se3tn_set_trunk_winograd(eng._h, -1, 80) == -1 and lib.se3tn_set_trunk_winograd(eng._h, 8, 101) == -1
    assert eng.get_trunk_winograd() == (16, 0)


def test_model_points_and_object_width(se3, tmp_path):
    rng = np.random.default_rng(1)
    pts = rng.uniform(-0.05, 0.05, (500, 3))
    ply = tmp_path / "m.ply"
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 500\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for p in pts:
            f.write("%.9f %.9f %.9f\n" % tuple(p))
    U = se3.utils
    got = U.load_model_points(str(ply))
    assert np.abs(got - pts).max() < 1e-8
    ds = U.voxel_down_sample(got, 0.005)
    # (`property float` columns are float32, as plyfile / trimesh return them: 4e-9 at this magnitude)
    assert len(ds) <= 500 and np.all(ds.min(0) >= pts.min(0) - 1e-8) and np.all(ds.max(0) <= pts.max(0) + 1e-8)
    w = U.compute_obj_max_width(ds)
    from scipy.spatial.distance import pdist
    assert abs(w - pdist(ds).max() * 1000) < 1e-6
    assert U.crop_window(np.array([[48, 213], [381, 213], [48, 546], [381, 546]])) == (213, 48, 546, 381)


def test_winograd_tile_codes_offset_rule_and_the_ctypes_mirror(se3):
    """The constants the Python mirror repeats are the header's; se3tn_set_winograd / se3tn_set_offset_rule accept exactly the
    documented codes (host-only context: no device work)."""
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "se3tracknet.h")).read()

    def h(name, conv=int):
        return conv(re.search(r"#define %s ([0-9.]+)" % name, hdr).group(1))
    L = se3._lib
    assert (h("SE3TN_WINOGRAD_TILE_AUTO"), h("SE3TN_WINOGRAD_TILE_6_4"), h("SE3TN_WINOGRAD_TILE6_MIN_BATCH")) == \
        (L.WINOGRAD_TILE_AUTO, L.WINOGRAD_TILE_6_4, L.WINOGRAD_TILE6_MIN_BATCH)
    assert h("SE3TN_WINOGRAD_HEADS_TILE6_MAX_ROT", float) == L.WINOGRAD_HEADS_TILE6_MAX_ROT
    assert (h("SE3TN_OFFSET_RULE_NUMPY1"), h("SE3TN_OFFSET_RULE_NUMPY2")) == (L.OFFSET_RULE_NUMPY1, L.OFFSET_RULE_NUMPY2)
    eng = se3.Engine(device=-1, max_batch=1)
    lib = L.load()
    assert eng.get_winograd() == (h("SE3TN_WINOGRAD_DEFAULT_MIN_BATCH"), L.WINOGRAD_TILE_AUTO)
    for tile in (2, 4, 6, L.WINOGRAD_TILE_6_4, L.WINOGRAD_TILE_AUTO):
        eng.set_winograd(9, tile)
        assert eng.get_winograd() == (9, tile)
    eng.set_winograd(3)                                  # tile 0 keeps the tile
    assert eng.get_winograd() == (3, L.WINOGRAD_TILE_AUTO)
    for bad in (1, 3, 5, 8, 46 + 1, 63):
        assert lib.se3tn_set_winograd(eng._h, 6, bad) == -1
    assert eng.get_offset_rule() == "numpy1"             # the reference's pinned NumPy generation is the default
    eng.set_offset_rule("numpy2")
    assert eng.get_offset_rule() == "numpy2" and lib.se3tn_get_offset_rule(eng._h) == L.OFFSET_RULE_NUMPY2
    eng.set_offset_rule(L.OFFSET_RULE_NUMPY1)
    assert eng.get_offset_rule() == "numpy1"
    assert lib.se3tn_set_offset_rule(eng._h, 2) == -1 and lib.se3tn_get_offset_rule(None) == -1
