"""GPU: edge cases of the hot path, through the C ABI, against the CPU oracle.

Covers what the domain offers as "ragged / empty / maximum / null" inputs: crop windows that
leave the frame on every side or miss it entirely, all-invalid depth, up- and down-sampling crops,
odd frame sizes, negative-z (GL) poses, external NHWC inputs, batch == max_batch, samples > 1,
and argument errors."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as Fx
from oracle import se3_oracle as O


@pytest.fixture(scope="module")
def se3():
    import se3tracknet_amd
    return se3tracknet_amd


def _oracle_crop(rgb, depth, win, z_mm, mean, std, stats):
    """oracle for one se3tn_crop: crop window (l,t,r,b) -> normalised [4,176,176]."""
    l, t, r, b = win
    bb = np.array([[t, l], [b, l], [t, r], [b, r]], np.int32)
    H, W = depth.shape
    if r <= 0 or b <= 0 or l >= W or t >= H:  # window misses the frame: the reference's slicing would
        rgbc = np.zeros((176, 176, 3), np.uint8); dc = np.zeros((176, 176), np.uint16)  # raise; zeros here
    else:
        rgbc, dc = O.crop_bbox(rgb, depth, bb, (176, 176))
    P = np.eye(4); P[2, 3] = z_mm / 1000.0
    d = O.normalize_depth(dc, P)
    return O.normalize_channels(rgbc.astype(np.float32), d, mean[4 * stats:4 * stats + 4], std[4 * stats:4 * stats + 4])


WINDOWS = [
    ("inside", (213, 48, 546, 381)),
    ("left_top_out", (-120, -90, 200, 230)),
    ("right_bottom_out", (500, 300, 900, 700)),
    ("all_sides_out", (-50, -60, 700, 560)),
    ("tiny_upsample", (300, 200, 317, 217)),
    ("one_pixel", (100, 100, 101, 101)),
    ("non_square", (10, 20, 400, 150)),
    ("entirely_outside", (700, 500, 900, 700)),
    ("huge", (-2000, -2000, 2600, 2500)),
]


@pytest.mark.parametrize("name,win", WINDOWS, ids=[w[0] for w in WINDOWS])
def test_preprocess_windows_bit_exact(se3, name, win):
    rng = np.random.default_rng(5)
    H, W = 480, 640
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    depth = rng.integers(0, 2500, (H, W)).astype(np.uint16)
    mean, std = Fx.mean_std(3)
    eng = se3.Engine(0, 1)
    eng.set_normalization(mean, std)
    out = torch.empty((1, 176, 176, 4), device="cuda")
    for z_mm, stats in ((812.5, 1), (-640.25, 0)):  # positive (cv) and negative (gl) poseA z
        eng.preprocess([dict(rgb=torch.from_numpy(rgb).cuda(), depth=torch.from_numpy(depth.view(np.int16)).cuda(),
                             window=win, z_offset_mm=z_mm, stats=stats)], out)
        got = out[0].permute(2, 0, 1).cpu().numpy()
        want = _oracle_crop(rgb, depth, win, z_mm, mean, std, stats)
        assert (got == want).all(), (name, float(np.abs(got - want).max()))


def test_preprocess_odd_frame_and_all_invalid_depth(se3):
    rng = np.random.default_rng(9)
    H, W = 97, 131
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mean, std = Fx.mean_std(4)
    eng = se3.Engine(0, 3)
    eng.set_normalization(mean, std)
    out = torch.empty((3, 176, 176, 4), device="cuda")
    depths = [np.zeros((H, W), np.uint16), np.full((H, W), 100, np.uint16), np.full((H, W), 65535, np.uint16)]
    crops = [dict(rgb=torch.from_numpy(rgb).cuda(), depth=torch.from_numpy(d.view(np.int16)).cuda(),
                  window=(-5, 3, 120, 90), z_offset_mm=700.0, stats=1) for d in depths]
    eng.preprocess(crops, out)
    for i, d in enumerate(depths):
        want = _oracle_crop(rgb, d, (-5, 3, 120, 90), 700.0, mean, std, 1)
        got = out[i].permute(2, 0, 1).cpu().numpy()
        assert (got == want).all()
        # every depth is invalid -> 2000 everywhere
        assert np.allclose(got[3], (2000.0 - mean[7]) / std[7])


def test_more_crops_than_one_launch_holds(se3):
    """se3tn_preprocess chunks the descriptors into kernel-argument blocks of 64 (CropArgs::MAX)."""
    rng = np.random.default_rng(2)
    rgb = torch.from_numpy(rng.integers(0, 256, (200, 260, 3), dtype=np.uint8)).cuda()
    depth = torch.from_numpy(rng.integers(300, 1500, (200, 260)).astype(np.uint16).view(np.int16)).cuda()
    eng = se3.Engine(0, 64)
    mean, std = Fx.mean_std(0)
    eng.set_normalization(mean, std)
    n = 133
    out = torch.empty((n, 176, 176, 4), device="cuda")
    wins = [(int(i * 3 - 20), int(10 - i % 60), int(i * 3 + 150), int(180 - i % 60)) for i in range(n)]
    eng.preprocess([dict(rgb=rgb, depth=depth, window=w, z_offset_mm=600.0 + i, stats=i & 1) for i, w in enumerate(wins)], out)
    r, d = rgb.cpu().numpy(), depth.cpu().numpy().view(np.uint16)
    for i in (0, 23, 24, 63, 64, 65, 127, 128, 132):
        want = _oracle_crop(r, d, wins[i], 600.0 + i, mean, std, i & 1)
        assert (out[i].permute(2, 0, 1).cpu().numpy() == want).all(), i


def test_pack_crops_descriptors_equal_the_dict_path(se3):
    """The vectorised descriptor builder (one tensor of frames, numpy record array) drives
    se3tn_preprocess to the same bits as per-crop dicts."""
    rng = np.random.default_rng(4)
    n = 30
    rgb = torch.from_numpy(rng.integers(0, 256, (n, 120, 160, 3), dtype=np.uint8)).cuda()
    depth = torch.from_numpy(rng.integers(300, 1500, (n, 120, 160)).astype(np.uint16).view(np.int16)).cuda()
    eng = se3.Engine(0, 32)
    mean, std = Fx.mean_std(0)
    eng.set_normalization(mean, std)
    wins = np.array([(i - 10, 5 - i, i + 100, 110 - i) for i in range(n)])
    z = 500.0 + np.arange(n)
    a = torch.empty((n, 176, 176, 4), device="cuda"); b = torch.empty_like(a)
    eng.preprocess([dict(rgb=rgb[i], depth=depth[i], window=tuple(wins[i]), z_offset_mm=z[i], stats=1) for i in range(n)], a)
    rec = se3.pack_crops(rgb, depth, wins, z, 1)
    assert rec.dtype.itemsize == 56 and rec.shape == (n,)
    eng.preprocess(rec, b)
    assert torch.equal(a, b)
    want = _oracle_crop(rgb[7].cpu().numpy(), depth[7].cpu().numpy().view(np.uint16), tuple(wins[7]), z[7], mean, std, 1)
    assert (b[7].permute(2, 0, 1).cpu().numpy() == want).all()


def test_external_nhwc_equals_nchw_and_internal_buffers(se3):
    sd = O.make_state_dict(0)
    eng = se3.Engine(0, 4)
    eng.load_state_dict(sd)
    A, B = Fx.net_inputs(21, 4)
    t1 = torch.empty((4, 3), device="cuda"); r1 = torch.empty((4, 3), device="cuda")
    t2 = torch.empty_like(t1); r2 = torch.empty_like(r1)
    eng.infer(A.cuda(), B.cuda(), 4, se3.NCHW, t1, r1)
    eng.infer(A.permute(0, 2, 3, 1).contiguous().cuda(), B.permute(0, 2, 3, 1).contiguous().cuda(), 4, se3.NHWC, t2, r2)
    assert (t1 == t2).all() and (r1 == r2).all()
    ref = O.forward(sd, A, B)
    assert float((t1.cpu() - ref["trans"]).abs().max()) < 1e-4


def test_batch_equals_max_batch_and_over(se3):
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=3)
    m.load_state_dict(sd); m.cuda(0)
    A, B = Fx.net_inputs(8, 4)
    out = m(A[:3].cuda(), B[:3].cuda())
    ref = O.forward(sd, A[:3], B[:3])
    assert float((out["rot"].cpu() - ref["rot"]).abs().max()) < 1e-4
    assert float((out["feature"].cpu() - ref["feature"]).abs().max()) < 5e-6 * float(ref["feature"].abs().max())
    with pytest.raises(se3._lib.Se3tnError):  # n > max_batch is refused, not truncated
        m(A.cuda(), B.cuda())


def test_error_paths_on_device_context(se3):
    eng = se3.Engine(0, 1)
    A = torch.zeros((1, 4, 176, 176), device="cuda")
    with pytest.raises(se3._lib.Se3tnError, match="weights"):
        eng.infer(A, A, 1, se3.NCHW)          # no weights bound
    with pytest.raises(se3._lib.Se3tnError, match="normalization"):
        eng.preprocess([dict(rgb=torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda"),
                             depth=torch.zeros((4, 4), dtype=torch.int16, device="cuda"),
                             window=(0, 0, 4, 4), z_offset_mm=0.0, stats=0)], torch.empty((1, 176, 176, 4), device="cuda"))
    eng.set_normalization(np.zeros(8), np.ones(8))
    with pytest.raises(se3._lib.Se3tnError, match="crop"):  # empty window
        eng.preprocess([dict(rgb=torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda"),
                             depth=torch.zeros((4, 4), dtype=torch.int16, device="cuda"),
                             window=(2, 2, 2, 5), z_offset_mm=0.0, stats=0)], torch.empty((1, 176, 176, 4), device="cuda"))
    eng.load_state_dict(O.make_state_dict(0))
    lib = se3._lib.load()
    rc = lib.se3tn_infer(eng._h, C.c_void_p(A.data_ptr()), C.c_void_p(A.data_ptr()), 1, 0, None, None,
                         C.c_void_p(A.data_ptr()), None, None)
    assert rc == -1  # poseA without poseB
    blob = torch.zeros(eng.packed_bytes(), dtype=torch.uint8, device="cuda")
    with pytest.raises(se3._lib.Se3tnError, match="header"):  # a blob that is not ours
        eng.bind_blob(blob)


class _Render:
    def render(self, ob2cam, K, window):
        return Fx.synthetic_render(7, ob2cam[2, 3])


def test_tracker_samples_gt_1_and_bind_blob_path(se3):
    """samples>1: the reference evaluates identical hypotheses and returns the first
    (predict.py:229-231,294-296); weights arriving through the multi-GPU bind path give the same pose."""
    sd = O.make_state_dict(0, head_gain=0.002)
    mean, std = Fx.mean_std(0)
    trk = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, renderer=_Render(), max_samples=4)
    rgb, depth = Fx.synthetic_frame(40)
    P0 = Fx.pose(3)
    P1 = trk.on_track(P0, rgb, depth, samples=1)
    P2 = trk.on_track(P0, rgb, depth, samples=2)
    P4 = trk.on_track(P0, rgb, depth, samples=4)
    assert (P1 == P2).all()                      # one pair alone or two together: the same split-K partition, the same bits
    assert np.abs(P1 - P4).max() < 1e-7          # from 3 pairs on the K partition follows the grid: float32 rounding only
    want, _ = O.on_track(sd, P0, rgb, depth, *Fx.synthetic_render(7, P0[2, 3]), Fx.K_YCB, 250.0, mean, std)
    assert np.abs(P1 - want).max() < 1e-5
    # same weights through pack -> (device copy standing in for the RCCL broadcast) -> bind
    eng2 = se3.Engine(0, 1)
    blob = se3.Engine(-1, 1).pack_state_dict(sd).cuda()
    eng2.bind_blob(blob)
    A, B = Fx.net_inputs(4, 1)
    t_a = torch.empty((1, 3), device="cuda"); t_b = torch.empty((1, 3), device="cuda")
    eng2.infer(A.cuda(), B.cuda(), 1, se3.NCHW, t_a, None)
    trk.engine.infer(A.cuda(), B.cuda(), 1, se3.NCHW, t_b, None)
    assert (t_a == t_b).all()


def test_on_track_batch_equals_sequential(se3):
    sd = O.make_state_dict(0, head_gain=0.002)
    mean, std = Fx.mean_std(0)
    trk = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, renderer=_Render(), max_samples=8)
    frames = [Fx.synthetic_frame(70 + i) for i in range(5)]
    poses = [Fx.pose(3 + i, (0.02 * i - 0.04, 0.01, 0.7 + 0.05 * i)) for i in range(5)]
    seq = np.stack([trk.on_track(poses[i], *frames[i]) for i in range(5)])
    bat = trk.on_track_batch(poses, [f[0] for f in frames], [f[1] for f in frames])
    assert bat.shape == (5, 4, 4)
    assert np.abs(bat - seq).max() < 1e-7   # same kernels (small-batch path), fixed-order reductions; the split-K partition of a
                                            # 5-pair call differs from a 1-pair call's: float32 rounding (1e-8 class), deterministic
    bat2 = trk.on_track_batch(poses, [f[0] for f in frames], [f[1] for f in frames])
    assert (bat2 == bat).all()
    with pytest.raises(ValueError):
        trk.on_track_batch(poses * 2, [f[0] for f in frames] * 2, [f[1] for f in frames] * 2)


@pytest.mark.parametrize("n", [1, 3, 8, 21])
def test_on_track_batch_one_call_equals_the_stepwise_path(se3, n):
    """se3tn_on_track_batch (image A of all n poses in four batched rasteriser launches, windows staged in one copy, one library
    call) against the step-by-step path of the same Tracker (a render + two uploads per pair): bit-identical image A, bbox, (trans,
    rot) and pose for every pair -- windows inside the frame, crossing its border and missing it altogether, frames repeated."""
    sd = O.make_state_dict(0, head_gain=0.01)
    mean, std = Fx.mean_std(0)
    mesh = Fx.icosphere(3, 0.05, 2)
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=150.0), mean, std, {"state_dict": sd}, max_samples=n)
    trk.renderer = se3.HipRenderer(trk.engine, mesh)
    frames = [Fx.synthetic_frame(70 + i) for i in range(4)]
    ts = [(0.0, 0.0, 0.8), (0.21, 0.0, 0.7), (-0.2, -0.17, 0.75), (0.0, 0.15, 0.5), (0.6, 0.5, 0.9)]    # the last: window off the frame
    poses = [Fx.pose(3 + i, ts[i % len(ts)]) for i in range(n)]
    rgbs, deps = [frames[i % 4][0] for i in range(n)], [frames[i % 4][1] for i in range(n)]
    assert trk.one_call
    got = trk.on_track_batch(poses, rgbs, deps)
    lp = trk.last_prediction
    g = dict(trans=lp["trans"].copy(), rot=lp["rot"].copy(), bbox=lp["bbox"].copy(),
             rgbA=[t.cpu().numpy() for t in lp["rgbA"]], depthA=[t.cpu().numpy() for t in lp["depthA"]])
    trk.one_call = False
    want = trk.on_track_batch(poses, rgbs, deps)
    lw = trk.last_prediction
    assert got.shape == (n, 4, 4) and np.array_equal(got, want)
    assert np.array_equal(g["trans"], lw["trans"]) and np.array_equal(g["rot"], lw["rot"]) and np.array_equal(g["bbox"], lw["bbox"])
    for i in range(n):
        assert np.array_equal(g["rgbA"][i], lw["rgbA"][i].cpu().numpy()) and np.array_equal(g["depthA"][i], lw["depthA"][i].cpu().numpy()), i
        assert (g["depthA"][i] != 0).sum() > 500
    # and per pair the single-frame call (batch-1 kernels: other summation order, same tolerance class as on_track's own batch test)
    trk.one_call = True
    one = np.stack([trk.on_track(poses[i], rgbs[i], deps[i]) for i in range(min(n, 5))])
    assert np.abs(one - got[:len(one)]).max() < 1e-6


def test_on_track_refuses_poses_that_have_no_crop_window(se3):
    """se3tn_on_track / se3tn_on_track_batch reject a pose at or behind the camera plane, a non-finite pose and a
    non-finite projection BEFORE the float -> int32 cast of compute_bbox's corners (undefined behaviour in C++), with an error code
    and text -- and the context keeps working afterwards."""
    sd = O.make_state_dict(0, head_gain=0.01)
    mean, std = Fx.mean_std(0)
    trk = se3.Tracker(dict(Fx.DATASET_INFO, object_width=150.0), mean, std, {"state_dict": sd}, max_samples=3)
    trk.renderer = se3.HipRenderer(trk.engine, Fx.icosphere(2, 0.05, 1))
    rgb, depth = Fx.synthetic_frame(71)
    good = Fx.pose(3, (0.0, 0.0, 0.8))
    want = trk.on_track(good, rgb, depth)
    bad = []
    for z in (0.0, -0.5, float("nan"), float("inf"), 1e-12):
        P = good.copy(); P[2, 3] = z
        bad.append(P)
    P = good.copy(); P[0, 3] = float("nan")
    bad.append(P)
    for P in bad:
        with pytest.raises(se3._lib.Se3tnError, match="camera|window"):
            trk.on_track(P, rgb, depth)
        with pytest.raises(se3._lib.Se3tnError, match="camera|window"):
            trk.on_track_batch([good, P, good], [rgb] * 3, [depth] * 3)
    assert np.array_equal(trk.on_track(good, rgb, depth), want)                         # the context is intact
    assert np.array_equal(trk.on_track_batch([good] * 3, [rgb] * 3, [depth] * 3)[1], want)
    with pytest.raises(ValueError):                                                     # frames of one call have one size
        trk.on_track_batch([good, good], [rgb, rgb[:100]], [depth, depth[:100]])


def test_hipgraph_replay_matches_eager(se3):
    """se3tn_enable_graphs: the captured se3tn_infer replays bit-identically, tracks argument changes
    (new pointers -> new capture) and content changes (same buffers, new data)."""
    sd = O.make_state_dict(0, head_gain=0.002)
    mean, std = Fx.mean_std(0)
    eager = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, renderer=_Render(), use_graphs=False)
    graph = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, renderer=_Render(), use_graphs=True)
    Pe = Pg = Fx.pose(3)
    for f in range(5):  # call 1 eager, call 2 captures, calls 3.. replay
        rgb, depth = Fx.synthetic_frame(90 + f)
        Pe = eager.on_track(Pe, rgb, depth)
        Pg = graph.on_track(Pg, rgb, depth)
        assert (Pe == Pg).all(), f
    assert any(g for g in [graph.engine]) and graph._stream is not None


@pytest.mark.parametrize("n", [8, 64])
def test_hipgraph_replay_of_the_winograd_path(se3, n):
    """n = 8 >= SE3TN_WINOGRAD_DEFAULT_MIN_BATCH: the captured graph contains the Winograd transform and GEMM kernels (F(4x4) fused
    blocks); n = 64: the F(6x6) fused blocks with the persistent 128 x 256 GEMM and the fused trunk kernel.  Replays are bit-identical
    to eager and follow new input data in the same buffers."""
    sd = O.make_state_dict(0)
    eng = se3.Engine(0, n)
    eng.load_state_dict(sd)
    assert eng.get_winograd()[0] <= n
    tr, ro = torch.empty((n, 3), device="cuda"), torch.empty((n, 3), device="cuda")
    A, B = Fx.net_inputs(21, n)
    Ac, Bc = A.cuda(), B.cuda()
    eng.infer(Ac, Bc, n, se3.NCHW, tr, ro)
    want = (tr.clone(), ro.clone(), eng.logits(n).clone())
    torch.cuda.synchronize()     # the context's workspaces are about to be used from another stream: one stream at a time per context
    st = torch.cuda.Stream()
    eng.enable_graphs(True)
    with torch.cuda.stream(st):
        for it in range(4):          # eager, capture, replay, replay
            tr.zero_(); ro.zero_()
            eng.infer(Ac, Bc, n, se3.NCHW, tr, ro)
            st.synchronize()
            assert torch.equal(tr, want[0]) and torch.equal(ro, want[1]) and torch.equal(eng.logits(n), want[2]), it
        A2, B2 = Fx.net_inputs(22, n)
        Ac.copy_(A2.cuda()); Bc.copy_(B2.cuda())     # same pointers, new content
        eng.infer(Ac, Bc, n, se3.NCHW, tr, ro)
        st.synchronize()
        got = tr.clone()
    eng.enable_graphs(False)
    eng.infer(Ac, Bc, n, se3.NCHW, tr, ro)
    torch.cuda.synchronize()
    assert torch.equal(got, tr) and not torch.equal(got, want[0])


def test_hipgraph_follows_the_small_kernel_switch(se3):
    """the graph key holds every switch that selects a kernel family.  With graphs on, se3tn_set_small_kernels(0)
    after a capture must run the general kernels (bit-equal to eager with the switch off), not replay the batch-1 family's graph."""
    sd = O.make_state_dict(0)
    n = 2
    eng = se3.Engine(0, n)
    eng.load_state_dict(sd)
    tr, ro = torch.empty((n, 3), device="cuda"), torch.empty((n, 3), device="cuda")
    A, B = Fx.net_inputs(23, n)
    Ac, Bc = A.cuda(), B.cuda()
    want = {}
    for on in (True, False):                       # eager results of both families
        eng.set_small_kernels(on)
        eng.infer(Ac, Bc, n, se3.NCHW, tr, ro)
        want[on] = eng.logits(n).clone()
    torch.cuda.synchronize()
    assert not torch.equal(want[True], want[False])          # the two families differ in summation order
    st = torch.cuda.Stream()
    eng.enable_graphs(True)
    with torch.cuda.stream(st):
        for on in (True, False, True, False):
            eng.set_small_kernels(on)
            for it in range(3):                    # eager, capture, replay -- under this setting's own key
                eng.infer(Ac, Bc, n, se3.NCHW, tr, ro)
                st.synchronize()
                assert torch.equal(eng.logits(n), want[on]), (on, it)
    eng.enable_graphs(False)
    eng.set_small_kernels(True)


def test_large_batch_192_matches_single_pairs(se3):
    """Maximum-size style check: a 192-pair call (3x BASELINE's batch; offsets beyond 2^31 bytes in the
    stem buffer) agrees pair-by-pair with single-pair calls, in both arithmetic modes."""
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=192)
    m.load_state_dict(sd); m.cuda(0)
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn((192, 4, 176, 176), generator=g, device="cuda")
    B = torch.randn((192, 4, 176, 176), generator=g, device="cuda")
    big = m(A, B, return_feature=False)
    t, r = big["trans"].clone(), big["rot"].clone()
    for i in (0, 97, 191):
        one = m(A[i:i + 1], B[i:i + 1], return_feature=False)
        assert float((one["trans"][0] - t[i]).abs().max()) < 5e-6 and float((one["rot"][0] - r[i]).abs().max()) < 5e-6
    ref = O.forward(sd, A[[191]].cpu(), B[[191]].cpu())
    assert float((t[191].cpu() - ref["trans"][0]).abs().max()) < 1e-4
    m.engine.set_precision(se3._lib.PREC_F16X3)
    fast = m(A, B, return_feature=False)
    assert not m.engine.overflow()
    assert float((fast["trans"] - t).abs().max()) < 2e-5 and float((fast["rot"] - r).abs().max()) < 2e-5


def test_pipelined_engine_lanes_share_weights_and_match_single_engine(se3):
    """PipelinedEngine: two lanes (contexts + streams) on one weight blob; batches alternate over the lanes; every
    lane's result is bit-identical to a plain Engine's."""
    import torch
    from oracle import se3_oracle as O
    sd = O.make_state_dict(5)
    n = 8
    A, B = Fx.net_inputs(41, 3 * n)
    ref = se3.Engine(0, n)
    ref.load_state_dict(sd)
    pe = se3.PipelinedEngine(0, n, depth=2)
    pe.load_state_dict(sd)
    pe.set_normalizers(0.03, 5 * np.pi / 180)
    assert pe.engines[0]._blob is pe.engines[1]._blob          # one device copy of the weights
    outs, want = [], []
    Ac, Bc = A.cuda(), B.cuda()
    torch.cuda.synchronize()
    for k in range(3):
        eng, stream = pe.next_lane()
        t = torch.empty((n, 3), device="cuda"); r = torch.empty((n, 3), device="cuda")
        with torch.cuda.stream(stream):
            eng.infer(Ac[k * n:(k + 1) * n], Bc[k * n:(k + 1) * n], n, se3.NCHW, t, r)
        outs.append((t, r))
    pe.synchronize()
    for k in range(3):
        t = torch.empty((n, 3), device="cuda"); r = torch.empty((n, 3), device="cuda")
        ref.infer(Ac[k * n:(k + 1) * n], Bc[k * n:(k + 1) * n], n, se3.NCHW, t, r)
        torch.cuda.synchronize()
        assert torch.equal(outs[k][0], t) and torch.equal(outs[k][1], r)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f32", "f16x3", "direct"])
@pytest.mark.parametrize("n", [64, 8])
def test_two_contexts_queued_on_two_streams_are_bit_identical_to_one(se3, precision, n):
    """Throughput mode under load: two contexts, two streams, the infers of a whole round queued back to back
    with no host synchronisation in between; every launch must reproduce the single-context logits bit for
    bit.  Regression test for profiles/EXPERIMENTS.md items 13: with two f16x3 contexts in flight the average-pool of
    tail_kernel was wrong in 40-85 % of the launches, because hipcc had emitted `v_pk_add_f32 ... op_sel:[0,1]
    op_sel_hi:[1,0]`, a form that returns wrong lanes 48-63 on gfx950 while another kernel issues
    v_mfma_f32_32x32x16_f16 on the same CU (scripts/probes/pk_opsel.hip; tests/test_isa_lint.py guards the build)."""
    import torch
    from oracle import se3_oracle as O
    sd = O.make_state_dict(0)
    A, B = Fx.net_inputs(5, n)
    Ac, Bc = A.cuda(), B.cuda()
    engines = []
    for _ in range(2):
        e = se3.Engine(0, n)
        e.load_state_dict(sd)
        if precision == "f16x3":
            e.set_precision(se3._lib.PREC_F16X3)
        elif precision == "direct":
            e.set_winograd(0)
        engines.append(e)
    t = torch.empty((n, 3), device="cuda"); r = torch.empty((n, 3), device="cuda")
    refs = []
    for e in engines:
        e.infer(Ac, Bc, n, se3.NCHW, t, r)
        torch.cuda.synchronize()
        refs.append(e.logits(n).clone())
    assert torch.equal(refs[0], refs[1])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    rounds, per_round = 4, 8
    outs = [(torch.empty((n, 3), device="cuda"), torch.empty((n, 3), device="cuda")) for _ in range(per_round)]
    bad = []
    for k in range(rounds):
        torch.cuda.synchronize()
        caps = []
        for i in range(per_round):
            with torch.cuda.stream(streams[i % 2]):
                engines[i % 2].infer(Ac, Bc, n, se3.NCHW, outs[i][0], outs[i][1])
                caps.append(engines[i % 2].logits(n))
        torch.cuda.synchronize()
        bad += [(k, i, int((lg != refs[0]).any(1).sum())) for i, lg in enumerate(caps) if not torch.equal(lg, refs[0])]
    assert not bad, "launches (round, index, wrong rows) that differ from the single-context result: %s" % bad


def test_tracker_init_with_a_ycb_size_mesh_is_bounded(se3, tmp_path):
    """a 262 k-vertex / 524 k-face binary PLY (the size of a YCB scan) through Tracker.__init__
    (predict.py:131-142, 180-182): vectorised loaders, bounded start-up, object_width as from a per-record parse."""
    import time
    from test_model_loaders import _big_mesh, _write_binary_ply   # tests/ is on sys.path (pytest rootdir conftest)
    v, n, c, f = _big_mesh()
    path = str(tmp_path / "ycb_size.ply")
    _write_binary_ply(path, v, n, c, f)
    info = {k: val for k, val in Fx.DATASET_INFO.items() if k != "object_width"}
    mean, std = Fx.mean_std(0)
    t0 = time.perf_counter()
    trk = se3.Tracker(info, mean, std, {"state_dict": O.make_state_dict(0)}, model_path=path)
    dt = time.perf_counter() - t0
    assert dt < 30.0, "Tracker.__init__ took %.1f s" % dt
    pts = v.astype(np.float64)
    w = se3.utils.compute_obj_max_width(se3.utils.voxel_down_sample(pts, 0.005))
    assert abs(trk.object_width - (w + info["boundingbox"] / 100 * w)) < 1e-9
    assert trk.renderer is not None and len(trk.renderer.mesh["faces"]) == len(f)
    rgbA, depthA = trk.render_window(Fx.pose(3))
    assert rgbA.shape == (176, 176, 3) and (depthA > 0).any()
