"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU
oracle and the committed goldens, on the same seeded inputs.

Tolerances (north star): |d(trans, rot)| <= 1e-4 on the network regression, <= 1e-5 on the composed
4x4 pose.  Pre-tanh logits and every intermediate feature map are compared too (tanh saturation
and the input-independent part of the logits would otherwise hide errors)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as Fx
from oracle import se3_oracle as O
from oracle.make_golden import LARGE_CASES, ON_TRACK_HEAD_GAIN, PRE_CASES, SUB

NET_TOL = 1e-4     # north-star tolerance on (trans, rot)
ACT_RTOL = 2e-5    # feature maps: f32 accumulation-order noise (oracle self-noise is ~2e-6)
POSE_TOL = 1e-5


@pytest.fixture(scope="module")
def se3():
    import se3tracknet_amd
    return se3tracknet_amd


@pytest.fixture(scope="module")
def model0(se3):
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=64)
    m.load_state_dict(sd)
    m.cuda(0).eval()
    return m, sd


def _nchw(t, border=0):  # (zero-bordered) NHWC cuda -> NCHW cpu interior
    if border:
        assert float(t[:, 0].abs().max()) == 0 and float(t[:, -1].abs().max()) == 0  # borders stay zero
        assert float(t[:, :, 0].abs().max()) == 0 and float(t[:, :, -1].abs().max()) == 0
        t = t[:, border:-border, border:-border]
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def _close(name, got, want, rtol, atol, scale_atol=0.0):
    """|got-want| <= atol + scale_atol*max|want| + rtol*|want| elementwise.  scale_atol covers
    f32 accumulation-order noise on values that are small differences of O(max|want|) terms."""
    got = got.double(); want = want.double()
    err = (got - want).abs()
    tol = atol + scale_atol * float(want.abs().max()) + rtol * want.abs()
    worst = float((err - tol).max())
    assert worst <= 0, "%s: max abs err %.3e (max |ref| %.3e), exceeds tol by %.3e" % (
        name, float(err.max()), float(want.abs().max()), worst)
    return float(err.max())


def test_single_hip_runtime(se3):
    torch.zeros(1, device="cuda")
    libs = {l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}
    assert len(libs) == 1, libs


def test_forward_every_stage_vs_oracle_and_golden(se3, model0, golden_dir):
    model, sd = model0
    g = np.load(os.path.join(golden_dir, "network_n3.npz"))
    A, B = Fx.net_inputs(1, 3)
    ref = O.forward(sd, A, B, intermediates=True)
    eng = model.engine
    n = 3
    # the stage maps: at 1-5 pairs the default kernels never WRITE the un-pooled stem map (stem_pool_small) nor the last head map
    # (tail_parts_kernel adds the partial-sum slices itself); se3tn_keep_intermediates makes every stage materialise
    eng.keep_intermediates(True)
    out = model(A.cuda(), B.cuda())
    torch.cuda.synchronize()
    stem = _nchw(eng.debug_buffer("stem", n))
    _close("stemA", stem[:, :64], ref["stemA"], ACT_RTOL, 1e-5)
    _close("stemB", stem[:, 64:], ref["stemB"], ACT_RTOL, 1e-5)
    pool = _nchw(eng.debug_buffer("pool", n), 1)
    _close("poolA", pool[:, :64], ref["poolA"], ACT_RTOL, 1e-5)
    cat = _nchw(eng.debug_buffer("q64", n), 1)
    _close("cat(a,b)", cat, ref["cat"], ACT_RTOL, 0, 5e-6)
    feat = out["feature"].cpu()
    _close("feature", feat, ref["feature"], ACT_RTOL, 0, 5e-6)
    head = _nchw(eng.debug_buffer("head", n), 1)
    _close("trans_conv2", head[:, :512], ref["trans_c2"], ACT_RTOL, 0, 5e-6)
    _close("rot_conv2", head[:, 512:], ref["rot_c2"], ACT_RTOL, 0, 5e-6)
    lg_kept = eng.logits(n).cpu()
    _close("trans_logit (stages kept)", lg_kept[:, :3], ref["trans_logit"], 0, NET_TOL)
    _close("rot_logit (stages kept)", lg_kept[:, 3:], ref["rot_logit"], 0, NET_TOL)
    # ... and the default configuration (the batch 1-5 kernel family end to end)
    eng.keep_intermediates(False)
    out = model(A.cuda(), B.cuda())
    torch.cuda.synchronize()
    assert float((out["feature"].cpu() - feat).abs().max()) <= 5e-5 * float(feat.abs().max())
    lg = eng.logits(n).cpu()
    _close("trans_logit", lg[:, :3], ref["trans_logit"], 0, NET_TOL)
    _close("rot_logit", lg[:, 3:], ref["rot_logit"], 0, NET_TOL)
    e1 = _close("trans", out["trans"].cpu(), ref["trans"], 0, NET_TOL)
    e2 = _close("rot", out["rot"].cpu(), ref["rot"], 0, NET_TOL)
    # and against what the reference's own code produced
    _close("trans vs golden", out["trans"].cpu(), torch.from_numpy(g["trans"]), 0, NET_TOL)
    _close("rot vs golden", out["rot"].cpu(), torch.from_numpy(g["rot"]), 0, NET_TOL)
    _close("feature vs golden", feat[:, ::SUB, ::SUB, ::SUB], torch.from_numpy(g["feature"]), ACT_RTOL, 0, 5e-6)
    _close("trans_conv2 vs golden", head[:, :512][:, ::SUB, ::SUB, ::SUB], torch.from_numpy(g["act_trans_conv2"]), ACT_RTOL, 0, 5e-6)
    print("max |d(trans,rot)| = %.2e" % max(e1, e2))


def test_forward_big_inputs_second_seed(se3, golden_dir):
    g = np.load(os.path.join(golden_dir, "network_big_n2.npz"))
    sd = O.make_state_dict(7, head_gain=0.002)
    m = se3.Se3TrackNet(176, max_batch=2)
    m.load_state_dict(sd)
    m.cuda(0)
    A, B = Fx.net_inputs(11, 2, scale=40.0)
    out = m(A.cuda(), B.cuda(), return_feature=False)
    lg = m.engine.logits(2).cpu()
    _close("logits", lg, torch.from_numpy(np.concatenate([g["trans_logit"], g["rot_logit"]], 1)), 0, NET_TOL)
    _close("trans", out["trans"].cpu(), torch.from_numpy(g["trans"]), 0, NET_TOL)
    _close("rot", out["rot"].cpu(), torch.from_numpy(g["rot"]), 0, NET_TOL)


def test_batch64_rows_equal_batch1_and_ragged_tail(se3, model0):
    """Size-independent property at BASELINE's batch: every pair of a batch-64 call (big-tile slab
    kernels) gives the same answer as that pair alone (split-K latency kernels; tiles straddle pair
    boundaries; 64*1936 etc. are not tile multiples), and n=5 (ragged last tile everywhere) and n=20
    (mixed: some layers split-K, some not) agree as well.  Different summation orders: f32 noise."""
    model, sd = model0
    A, B = Fx.net_inputs(5, 64)
    Ac, Bc = A.cuda(), B.cuda()
    o64 = model(Ac, Bc, return_feature=False)
    t64, r64 = o64["trans"].clone(), o64["rot"].clone()
    l64 = model.engine.logits(64).clone()
    for i in (0, 17, 63):
        o1 = model(Ac[i:i + 1], Bc[i:i + 1], return_feature=False)
        l1 = model.engine.logits(1)
        assert float((l1[0] - l64[i]).abs().max()) < 5e-6
        assert float((o1["trans"][0] - t64[i]).abs().max()) < 5e-6
    o5 = model(Ac[10:15], Bc[10:15], return_feature=False)
    assert float((o5["rot"] - r64[10:15]).abs().max()) < 5e-6
    o20 = model(Ac[30:50], Bc[30:50], return_feature=False)
    assert float((o20["trans"] - t64[30:50]).abs().max()) < 5e-6
    # the split-K path sums its slices in a fixed order: bit-reproducible run to run
    o1a = model(Ac[7:8], Bc[7:8], return_feature=False)["trans"].clone()
    o1b = model(Ac[7:8], Bc[7:8], return_feature=False)["trans"].clone()
    assert (o1a == o1b).all()
    # spot-check 2 of the 64 against the CPU oracle
    ref = O.forward(sd, A[[3, 40]], B[[3, 40]])
    _close("b64 trans", t64[[3, 40]].cpu(), ref["trans"], 0, NET_TOL)
    _close("b64 rot", r64[[3, 40]].cpu(), ref["rot"], 0, NET_TOL)


@pytest.mark.parametrize("tile", [2, 4, 6])
def test_winograd_path_vs_direct_and_oracle(se3, golden_dir, tile):
    """The large-batch algorithm of the 256/512-channel residual blocks (Winograd F(tile x tile,3x3),
    float32) forced on at small n: every stage against the oracle and the reference-made golden, and
    against the direct kernels on the same engine.  n=3 and n=5 make every GEMM row tile ragged;
    11x11 and 22x22 maps exercise the dropped rows / columns of the 12- and 24-wide tilings."""
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=8)
    m.load_state_dict(sd)
    m.cuda(0).eval()
    eng = m.engine
    eng.keep_intermediates(True)   # the fused F(4x4) blocks do not store ab_t / head_t / head otherwise
    g = np.load(os.path.join(golden_dir, "network_n3.npz"))
    A, B = Fx.net_inputs(1, 3)
    ref = O.forward(sd, A, B, intermediates=True)
    eng.set_winograd(0)
    od = m(A.cuda(), B.cuda())
    feat_d, head_d = od["feature"].cpu().clone(), _nchw(eng.debug_buffer("head", 3), 1).clone()
    lg_d = eng.logits(3).cpu().clone()
    eng.set_winograd(1, tile)
    ow = m(A.cuda(), B.cuda())
    feat_w, head_w = ow["feature"].cpu(), _nchw(eng.debug_buffer("head", 3), 1)   # _nchw: borders still zero
    lg_w = eng.logits(3).cpu()
    assert not torch.equal(head_w, head_d), "the Winograd path did not run"
    WINO_SCALE = {2: 2e-5, 4: 6e-5, 6: 1.5e-4}[tile]   # transform-amplified f32 rounding, relative to the layer's largest activation
    _close("feature", feat_w, ref["feature"], ACT_RTOL, 0, WINO_SCALE)
    _close("trans_conv2", head_w[:, :512], ref["trans_c2"], ACT_RTOL, 0, WINO_SCALE)
    _close("rot_conv2", head_w[:, 512:], ref["rot_c2"], ACT_RTOL, 0, WINO_SCALE)
    _close("feature vs direct", feat_w, feat_d, ACT_RTOL, 0, WINO_SCALE)
    _close("head vs direct", head_w, head_d, ACT_RTOL, 0, WINO_SCALE)
    e0 = _close("logits vs direct", lg_w, lg_d, 0, 2e-5)
    _close("trans_logit", lg_w[:, :3], ref["trans_logit"], 0, NET_TOL)
    _close("rot_logit", lg_w[:, 3:], ref["rot_logit"], 0, NET_TOL)
    _close("trans vs golden", ow["trans"].cpu(), torch.from_numpy(g["trans"]), 0, NET_TOL)
    _close("rot vs golden", ow["rot"].cpu(), torch.from_numpy(g["rot"]), 0, NET_TOL)
    _close("feature vs golden", feat_w[:, ::SUB, ::SUB, ::SUB], torch.from_numpy(g["feature"]), ACT_RTOL, 0, WINO_SCALE)
    # ragged n, and the zero borders of every buffer the output transform writes stay zero
    A5, B5 = Fx.net_inputs(9, 5)
    o5 = m(A5.cuda(), B5.cuda(), return_feature=False)
    t5, r5 = o5["trans"].cpu().clone(), o5["rot"].cpu().clone()
    for name in ("ab", "ab_t", "head", "head_t"):
        _nchw(eng.debug_buffer(name, 5), 1)
    ref5 = O.forward(sd, A5, B5)
    _close("n5 trans", t5, ref5["trans"], 0, NET_TOL)
    _close("n5 rot", r5, ref5["rot"], 0, NET_TOL)
    eng.set_winograd(0)
    o5d = m(A5.cuda(), B5.cuda(), return_feature=False)
    e1 = _close("n5 trans vs direct", t5, o5d["trans"].cpu(), 0, 2e-5)
    print("F(%dx%d): max |d logit| Winograd vs direct = %.2e, |d trans| = %.2e" % (tile, tile, e0, e1))


def test_fused_trunk_winograd_vs_direct_and_oracle(se3, golden_dir):
    """The large-batch algorithm of the 64-channel trunk (fused Winograd F(2x2,3x3), wino64_fused.hip) forced on at small n
    (min_fill 0): the trunk's buffers against the direct kernels on the same engine, the logits against the oracle and the
    reference-made golden; the zero borders of the padded maps stay zero; and the default rule picks it per launch (whole rounds of
    workgroups only)."""
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=64)
    m.load_state_dict(sd)
    m.cuda(0).eval()
    eng = m.engine
    g = np.load(os.path.join(golden_dir, "network_n3.npz"))
    A, B = Fx.net_inputs(1, 3)
    ref = O.forward(sd, A, B, intermediates=True)
    eng.set_trunk_winograd(0)
    od = m(A.cuda(), B.cuda())
    q_d, t_d = _nchw(eng.debug_buffer("q64", 3), 1).clone(), _nchw(eng.debug_buffer("t64", 3), 1).clone()
    lg_d, feat_d = eng.logits(3).cpu().clone(), od["feature"].cpu().clone()
    eng.set_trunk_winograd(1, 0)
    ow = m(A.cuda(), B.cuda())
    q_w, t_w = _nchw(eng.debug_buffer("q64", 3), 1), _nchw(eng.debug_buffer("t64", 3), 1)   # _nchw: borders still zero
    lg_w = eng.logits(3).cpu()
    assert not torch.equal(q_w, q_d), "the fused trunk path did not run"
    SCALE = 2e-5   # F(2x2) transform-amplified f32 rounding, relative to the layer's largest activation
    _close("q64 vs direct", q_w, q_d, ACT_RTOL, 0, SCALE)
    _close("t64 vs direct", t_w, t_d, ACT_RTOL, 0, SCALE)
    _close("feature", ow["feature"].cpu(), ref["feature"], ACT_RTOL, 0, SCALE)
    _close("feature vs direct", ow["feature"].cpu(), feat_d, ACT_RTOL, 0, SCALE)
    e0 = _close("logits vs direct", lg_w, lg_d, 0, 2e-5)
    _close("trans_logit", lg_w[:, :3], ref["trans_logit"], 0, NET_TOL)
    _close("rot_logit", lg_w[:, 3:], ref["rot_logit"], 0, NET_TOL)
    _close("trans vs golden", ow["trans"].cpu(), torch.from_numpy(g["trans"]), 0, NET_TOL)
    _close("rot vs golden", ow["rot"].cpu(), torch.from_numpy(g["rot"]), 0, NET_TOL)
    # every pair of a forced batch of 5 equals that pair alone through the same kernel (a workgroup never spans two images): bitwise.
    # (At 1-2 pairs the stem + pool take another kernel than at five -- switched off here so that one pair runs the same kernels as five.)
    eng.set_small_kernels(False)
    A5, B5 = Fx.net_inputs(9, 5)
    m(A5.cuda(), B5.cuda(), return_feature=False)
    q5 = eng.debug_buffer("q64", 5).clone()
    l5 = eng.logits(5).cpu().clone()
    m(A5[3:4].cuda(), B5[3:4].cuda(), return_feature=False)
    assert torch.equal(eng.debug_buffer("q64", 1)[0], q5[3]), "fused trunk: a pair's result depends on its batch"
    eng.set_small_kernels(True)
    ref5 = O.forward(sd, A5, B5)
    _close("n5 logits", l5, torch.cat([ref5["trans_logit"], ref5["rot_logit"]], 1), 0, NET_TOL)
    # the default rule (rounds of the 256 CUs at least 55 % full): n = 64 -> 2 | 1 rounds, n = 48 -> 1.5 | 0.75: all four launches;
    # n = 32 and n = 20: only the grouped A2|B2 launches (1.0 | 0.63 of a round; the single ones are 0.5 | 0.31); n = 16: none
    eng.set_trunk_winograd(se3._lib.TRUNK_WINOGRAD_DEFAULT_MIN_BATCH)
    A64, B64 = Fx.net_inputs(5, 64)
    Ac, Bc = A64.cuda(), B64.cuda()
    for n, want in ((64, 4), (48, 4), (32, 2), (20, 2), (16, 0), (4, 0)):
        eng.profile_enable(1)
        m(Ac[:n], Bc[:n], return_feature=False)
        torch.cuda.synchronize()
        names = [nm for nm, _ in eng.profile_launches(0) if nm.startswith("conv64")]
        eng.profile_enable(0)
        assert len(names) == 4 and sum("fused F(2x2)" in nm for nm in names) == want, (n, names)
    print("fused trunk F(2x2): max |d logit| vs direct = %.2e" % e0)


def _modes(se3, eng):
    """The three arithmetic configurations of the engine: (name, enter, leave)."""
    wmin, wtile = eng.get_winograd()
    tw = eng.get_trunk_winograd()
    return [
        ("f32 default (Winograd blocks from n >= %d, tile %s, fused F(2x2) trunk in full rounds)"
         % (wmin, "4 | 6 by batch size" if wtile == se3._lib.WINOGRAD_TILE_AUTO else wtile), lambda: None, lambda: None),
        ("f32 direct kernels only", lambda: (eng.set_winograd(0), eng.set_trunk_winograd(0)),
         lambda: (eng.set_winograd(wmin, wtile), eng.set_trunk_winograd(*tw))),
        ("f16x3", lambda: eng.set_precision(se3._lib.PREC_F16X3), lambda: eng.set_precision(se3._lib.PREC_F32)),
    ]


def _wino_launch_names(eng, run):
    """Names of the 256/512-channel conv launches of one profiled call (the library appends the algorithm: "[F(6x6)]")."""
    eng.profile_enable(1)
    try:
        run()
        torch.cuda.synchronize()
        return [nm for nm, _ in eng.profile_launches(0) if nm.startswith("convAB2") or nm.startswith("trans|rot conv2")]
    finally:
        eng.profile_enable(0)


LARGE_MAG_CASES = [c for c in LARGE_CASES if c[5] > 1.0]   # x40 / x150 inputs at n = 8 (AUTO -> F(4x4)) and n = 16 (AUTO -> F(6x6))


@pytest.mark.parametrize("case", LARGE_MAG_CASES, ids=[c[0] for c in LARGE_MAG_CASES])
def test_default_path_large_magnitude_inputs_vs_reference_golden(se3, golden_dir, case):
    """The engine's DEFAULT algorithm on inputs x40 / x150 (what real std.npy files produce) against logits made by the
    reference's own code: at n = 8 SE3TN_WINOGRAD_TILE_AUTO runs the fused Winograd F(4x4,3x3) blocks, at n = 16 (>=
    SE3TN_WINOGRAD_TILE6_MIN_BATCH) F(6x6,3x3) -- the algorithm BASELINE's batch of 64 runs -- and the launch names of a
    profiled call must say so.  F(6x6) is additionally FORCED at n = 8, F(4x4) at n = 16; the direct kernels and the f16x3
    mode are held to the same numbers."""
    fname, wseed, gain, iseed, n, scale = case
    g = np.load(os.path.join(golden_dir, fname + ".npz"))
    want = torch.from_numpy(np.concatenate([g["trans_logit"], g["rot_logit"]], 1))
    sd = O.make_state_dict(wseed, head_gain=gain)
    m = se3.Se3TrackNet(176, max_batch=n)
    m.load_state_dict(sd)
    m.cuda(0)
    eng = m.engine
    wmin, wtile = eng.get_winograd()
    assert wmin <= n, "the default threshold moved: regenerate the golden at a larger n"
    assert wtile == se3._lib.WINOGRAD_TILE_AUTO, "the default tile selection moved: revisit which n runs which algorithm"
    A, B = Fx.net_inputs(iseed, n, scale=scale)
    Ac, Bc = A.cuda(), B.cuda()
    auto_tile = 6 if n >= se3._lib.WINOGRAD_TILE6_MIN_BATCH else 4
    names = _wino_launch_names(eng, lambda: m(Ac, Bc, return_feature=False))
    assert len(names) == 4 and all("[F(%dx%d)]" % (auto_tile, auto_tile) in nm for nm in names), names
    other = 4 if auto_tile == 6 else 6
    modes = _modes(se3, eng)
    modes.insert(1, ("f32 with F(%dx%d) forced" % (other, other), lambda: eng.set_winograd(wmin, other), lambda: eng.set_winograd(wmin, wtile)))
    seen = {}
    for name, enter, leave in modes:
        enter()
        try:
            if "forced" in name:
                forced = _wino_launch_names(eng, lambda: m(Ac, Bc, return_feature=False))
                assert len(forced) == 4 and all("[F(%dx%d)]" % (other, other) in nm for nm in forced), forced
            out = m(Ac, Bc, return_feature=False)
            lg = eng.logits(n).cpu()
            if name == "f16x3" and eng.overflow():
                # documented behaviour: activations beyond the f16 range raise the flag and the caller reruns
                # in float32 -- x150 inputs may do that; silently wrong numbers are what must not happen
                print("%s: %s range guard fired (caller falls back to f32)" % (fname, name))
                continue
            e = _close(name + " logits vs reference golden", lg, want, 0, NET_TOL)
            _close(name + " trans", out["trans"].cpu(), torch.from_numpy(g["trans"]), 0, NET_TOL)
            _close(name + " rot", out["rot"].cpu(), torch.from_numpy(g["rot"]), 0, NET_TOL)
            print("%s  %s: max |d logit| vs reference = %.2e" % (fname, name, e))
            seen[name] = lg
        finally:
            leave()
    lgs = list(seen.values())
    assert not torch.equal(lgs[0], lgs[1]) and not torch.equal(lgs[0], lgs[2]) and not torch.equal(lgs[1], lgs[2]), \
        "default, forced-tile and direct-only runs must be three different algorithms"


def test_auto_tile_keeps_the_heads_on_f4x4_under_a_large_rot_normalizer(se3):
    """SE3TN_WINOGRAD_TILE_AUTO at n >= 14: F(6x6) for the 256-channel block always; for the 512-channel heads only while
    rot_normalizer <= SE3TN_WINOGRAD_HEADS_TILE6_MAX_ROT (their rounding reaches the composed pose x rot_normalizer).  Tile 6 forces
    F(6x6) everywhere, SE3TN_WINOGRAD_TILE_6_4 the mixed form whatever the normaliser; below 14 pairs AUTO is F(4x4)."""
    n = 16
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=n)
    m.load_state_dict(sd)
    m.cuda(0)
    eng = m.engine
    A, B = Fx.net_inputs(3, n)
    Ac, Bc = A.cuda(), B.cuda()
    ref = O.forward(sd, A[:2], B[:2])
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)

    def tiles(nn=n):
        names = _wino_launch_names(eng, lambda: m(Ac[:nn], Bc[:nn], return_feature=False))
        _close("logits", eng.logits(2).cpu(), want, 0, NET_TOL)
        return ["[F(6x6)]" in nm and 6 or "[F(4x4)]" in nm and 4 or 0 for nm in names]
    assert tiles() == [6, 6, 6, 6]                                  # default normalisers (0.03 m, 5 degrees)
    eng.set_normalizers(0.03, 30 * np.pi / 180)                     # predict.py:586
    assert tiles() == [6, 6, 4, 4]
    assert tiles(8) == [4, 4, 4, 4]
    eng.set_winograd(6, 6)
    assert tiles() == [6, 6, 6, 6]
    eng.set_normalizers(0.03, 5 * np.pi / 180)
    eng.set_winograd(6, se3._lib.WINOGRAD_TILE_6_4)
    assert tiles() == [6, 6, 4, 4]
    assert abs(se3._lib.WINOGRAD_HEADS_TILE6_MAX_ROT - 0.2) < 1e-12
    eng.set_winograd(6, se3._lib.WINOGRAD_TILE_AUTO)
    assert tiles() == [6, 6, 6, 6]


def test_reloading_weights_rederives_every_winograd_plane_set(se3):
    """a second load_state_dict on the SAME context re-uses the blob's device address; every derived plane set
    (F(4x4), F(6x6), the fused trunk's F(2x2), the f16x3 split panels) must follow the new weights.  n = 16 runs F(6x6) + the
    grouped fused-trunk launches (AUTO); the reloaded engine must equal a fresh engine bit for bit and the oracle within tolerance."""
    n = 20
    A, B = Fx.net_inputs(41, n)
    Ac, Bc = A.cuda(), B.cuda()
    sd1, sd2 = O.make_state_dict(0), O.make_state_dict(3, head_gain=0.05)
    m = se3.Se3TrackNet(176, max_batch=n)
    m.load_state_dict(sd1)
    m.cuda(0)
    names = _wino_launch_names(m.engine, lambda: m(Ac, Bc, return_feature=False))
    assert all("[F(6x6)]" in nm for nm in names), names
    l1 = m.engine.logits(n).cpu().clone()
    m.load_state_dict(sd2)                      # same context, same blob address, new contents
    m(Ac, Bc, return_feature=False)
    l2 = m.engine.logits(n).cpu().clone()
    fresh = se3.Se3TrackNet(176, max_batch=n)
    fresh.load_state_dict(sd2)
    fresh.cuda(0)
    fresh(Ac, Bc, return_feature=False)
    lf = fresh.engine.logits(n).cpu()
    assert torch.equal(l2, lf), "stale Winograd planes after a reload: max |d logit| %.3e" % float((l2 - lf).abs().max())
    assert not torch.equal(l1, l2)
    ref = O.forward(sd2, A[:3], B[:3])
    _close("reloaded logits vs oracle", l2[:3], torch.cat([ref["trans_logit"], ref["rot_logit"]], 1), 0, NET_TOL)
    # and every forced tile after the reload (F(2x2) / F(4x4) re-derive wino_u for the tile they are asked for)
    for tile in (2, 4, 6):
        m.engine.set_winograd(1, tile); fresh.engine.set_winograd(1, tile)
        m(Ac, Bc, return_feature=False); fresh(Ac, Bc, return_feature=False)
        assert torch.equal(m.engine.logits(n), fresh.engine.logits(n)), tile


def test_batch64_every_pair_vs_oracle_and_reference_golden(se3, model0, golden_dir):
    """BASELINE configs[1]'s batch: ALL 64 pairs of one call against the CPU oracle (run here on the same
    inputs) and against logits the reference's own code produced (tests/golden/network_n64.npz), in the
    default float32 configuration (Winograd blocks), with the direct kernels only, and in f16x3 mode."""
    model, sd = model0
    eng = model.engine
    fname, wseed, gain, iseed, n, scale = LARGE_CASES[2]
    assert (wseed, gain, n) == (0, 0.05, 64)
    g = np.load(os.path.join(golden_dir, fname + ".npz"))
    A, B = Fx.net_inputs(iseed, n, scale=scale)
    ref = O.forward(sd, A, B)
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
    gold = torch.from_numpy(np.concatenate([g["trans_logit"], g["rot_logit"]], 1))
    Ac, Bc = A.cuda(), B.cuda()
    logits = []
    for name, enter, leave in _modes(se3, eng):
        enter()
        try:
            out = model(Ac, Bc, return_feature=False)
            lg = eng.logits(n).cpu()
            assert not eng.overflow()
            e1 = _close(name + ": 64 logits vs oracle", lg, want, 0, NET_TOL)
            e2 = _close(name + ": 64 logits vs reference golden", lg, gold, 0, NET_TOL)
            for k in ("trans", "rot"):
                _close(name + ": " + k + " vs oracle", out[k].cpu(), ref[k], 0, NET_TOL)
                _close(name + ": " + k + " vs reference golden", out[k].cpu(), torch.from_numpy(g[k]), 0, NET_TOL)
            print("batch 64, %s: max |d logit| vs oracle %.2e, vs reference golden %.2e" % (name, e1, e2))
            logits.append(lg)
        finally:
            leave()
    assert not torch.equal(logits[0], logits[1]) and not torch.equal(logits[1], logits[2])


@pytest.mark.parametrize("tile", [4, 6])
def test_fused_winograd_blocks_keep_intermediates_is_bit_neutral(se3, model0, tile):
    """The fused F(4x4) / F(6x6) blocks (mid transform in LDS, tail reduced in registers) with and without the optional
    activation stores: identical bits; with the stores the kept tensors equal what the unfused per-conv path
    (F(2x2) selects it) would have left within Winograd rounding, and the zero borders stay zero."""
    model, sd = model0
    eng = model.engine
    wmin, wtile = eng.get_winograd()
    eng.set_winograd(wmin, tile)
    A, B = Fx.net_inputs(23, 16)
    Ac, Bc = A.cuda(), B.cuda()
    o0 = model(Ac, Bc)
    l0, f0, t0 = eng.logits(16).clone(), o0["feature"].clone(), o0["trans"].clone()
    eng.keep_intermediates(True)
    try:
        o1 = model(Ac, Bc)
        assert torch.equal(eng.logits(16), l0) and torch.equal(o1["feature"], f0) and torch.equal(o1["trans"], t0)
        head = _nchw(eng.debug_buffer("head", 16), 1)
        head_t = _nchw(eng.debug_buffer("head_t", 16), 1)
        ab_t = _nchw(eng.debug_buffer("ab_t", 16), 1)
        ref = O.forward(sd, A[:2], B[:2], intermediates=True)
        _close("kept trans_conv2", head[:2, :512], ref["trans_c2"], ACT_RTOL, 0, 6e-5 if tile == 4 else 1.5e-4)
        _close("kept rot_conv2", head[:2, 512:], ref["rot_c2"], ACT_RTOL, 0, 6e-5 if tile == 4 else 1.5e-4)
        assert float(head_t.abs().max()) > 0 and float(ab_t.abs().max()) > 0
        # the logits are the FC of the mean of the kept activation
        mean = head.mean(dim=(2, 3))
        lg = torch.cat([mean[:, :512] @ sd["trans_out.0.weight"].T + sd["trans_out.0.bias"],
                        mean[:, 512:] @ sd["rot_out.0.weight"].T + sd["rot_out.0.bias"]], 1)
        _close("logits from kept head", l0.cpu(), lg, 0, 2e-6)
    finally:
        eng.keep_intermediates(False)
        eng.set_winograd(wmin, wtile)


@pytest.mark.parametrize("tile", [4, 6])
def test_fused_winograd_blocks_equal_the_conv_by_conv_form(se3, tile):
    """One residual block as ONE launch sequence (in-transform, GEMM, [out | in] mid transform through LDS, GEMM, out-transform)
    against the same Winograd tile run conv by conv (SE3TN_WINOGRAD_FUSE=0: three launches per convolution, the intermediate
    activation through memory): the mid transform does the same operations in the same order, so `feature` (the output of the
    256-channel block) is BIT-identical; the heads' fused tail reduces the average pool per tile first, so the logits agree to
    float32 rounding only.  n = 20: AUTO's F(6x6) range, ragged GEMM row tiles."""
    n = 20
    sd = O.make_state_dict(0)
    A, B = Fx.net_inputs(61, n)
    Ac, Bc = A.cuda(), B.cuda()
    res = {}
    for fuse in ("1", "0"):
        os.environ["SE3TN_WINOGRAD_FUSE"] = fuse          # read by se3tn_create (the context is created by .cuda())
        try:
            m = se3.Se3TrackNet(176, max_batch=n)
            m.load_state_dict(sd)
            m.cuda(0)
        finally:
            del os.environ["SE3TN_WINOGRAD_FUSE"]
        m.engine.set_winograd(1, tile)
        names = _wino_launch_names(m.engine, lambda: m(Ac, Bc))
        assert len(names) == 4 and all(("[F(%dx%d)]" % (tile, tile)) in nm for nm in names), names
        assert all(("fused block" in nm) == (fuse == "1") for nm in names), names
        out = m(Ac, Bc)
        res[fuse] = (out["feature"].cpu().clone(), m.engine.logits(n).cpu().clone())
    assert torch.equal(res["1"][0], res["0"][0]), "fused mid transform != out-transform + in-transform"
    e = float((res["1"][1] - res["0"][1]).abs().max())
    assert 0 < e < 2e-6, e
    ref = O.forward(sd, A[:2], B[:2])
    _close("fused logits vs oracle", res["1"][1][:2], torch.cat([ref["trans_logit"], ref["rot_logit"]], 1), 0, NET_TOL)


@pytest.mark.parametrize("n", [20, 64])
def test_persistent_winograd_gemm_equals_the_tiled_one_bitwise(se3, n):
    """`wino_gemmp_kernel` (128 x 256 tiles, 8 waves, one persistent workgroup per CU, XCD-local tile order; the default from two
    tiles per CU = batch 64) against `wino_gemm_kernel` (128 | 96 x 128, 4 waves): same fragment layout and k order, so every
    per-frequency product -- and with them the logits and `feature` -- must be BIT-identical.  SE3TN_WINO_GEMMP = 1 | 0 forces
    either (read at se3tn_create).  n = 20: ragged row tiles (320 / 80 Winograd tiles: 3 / 1 tiles of 128 rows), fewer tiles than
    CUs x 2 (the persistent loop runs 1-2 tiles per workgroup); n = 64: exactly 2 tiles per CU."""
    sd = O.make_state_dict(0)
    A, B = Fx.net_inputs(77, n)
    Ac, Bc = A.cuda(), B.cuda()
    res = {}
    for mode in ("1", "0"):
        os.environ["SE3TN_WINO_GEMMP"] = mode
        try:
            m = se3.Se3TrackNet(176, max_batch=n)
            m.load_state_dict(sd)
            m.cuda(0)
        finally:
            del os.environ["SE3TN_WINO_GEMMP"]
        m.engine.set_winograd(1, 6)
        out = m(Ac, Bc)
        res[mode] = (out["feature"].cpu().clone(), m.engine.logits(n).cpu().clone())
    assert torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])
    ref = O.forward(sd, A[:2], B[:2])
    _close("logits vs oracle", res["1"][1][:2], torch.cat([ref["trans_logit"], ref["rot_logit"]], 1), 0, NET_TOL)


def test_batch_permutation_equivariance_bitwise(se3, model0):
    """Size-independent property at BASELINE's batch: permuting the 64 pairs permutes the outputs BITWISE
    (no pair's result depends on its neighbours or on where its pixels / Winograd tiles fall in a
    workgroup tile: every kernel accumulates a row's K dimension in a fixed order)."""
    model, _ = model0
    A, B = Fx.net_inputs(31, 64)
    Ac, Bc = A.cuda(), B.cuda()
    model(Ac, Bc, return_feature=False)
    l0 = model.engine.logits(64).clone()
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(5)).cuda()
    model(Ac[perm].contiguous(), Bc[perm].contiguous(), return_feature=False)
    l1 = model.engine.logits(64).clone()
    assert torch.equal(l1, l0[perm])
    assert not torch.equal(l0[0], l0[1])
    model(Ac, Bc, return_feature=False)                 # and run to run: no atomics anywhere on the path
    assert torch.equal(model.engine.logits(64), l0)


def _frame_to_cuda(rgb, depth):
    return torch.from_numpy(rgb).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()


@pytest.mark.parametrize("case", PRE_CASES, ids=[c[0] for c in PRE_CASES])
def test_preprocess_vs_oracle_and_golden(se3, case, golden_dir):
    name, fseed, t, width = case
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    eng = se3.Engine(0, 2)
    mean, std = Fx.mean_std(0)
    eng.set_normalization(mean, std)
    rgb, depth = Fx.synthetic_frame(fseed)
    P = Fx.pose(fseed, t)
    rgbA, depthA = Fx.synthetic_render(fseed + 100, t[2])
    bb = se3.compute_bbox(P, Fx.K_YCB, width)
    assert (bb == g[name + "_bbox"]).all()
    r_d, d_d = _frame_to_cuda(rgb, depth)
    ra_d, da_d = _frame_to_cuda(rgbA, depthA)
    out = torch.empty((2, 176, 176, 4), dtype=torch.float32, device="cuda")
    z = float(P[2, 3]) * 1000
    eng.preprocess([dict(rgb=ra_d, depth=da_d, window=(0, 0, 176, 176), z_offset_mm=z, stats=0),
                    dict(rgb=r_d, depth=d_d, window=se3.crop_window(bb), z_offset_mm=z, stats=1)], out)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).contiguous().cpu().numpy()
    rgbB, depthB = O.crop_bbox(rgb, depth, bb, (176, 176))
    a, b = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std)
    # byte/index work + IEEE float64 arithmetic: bit-exact
    assert (got[0] == a).all(), float(np.abs(got[0] - a).max())
    assert (got[1] == b).all(), float(np.abs(got[1] - b).max())
    assert Fx.sha(got[0]) == str(g[name + "_dataA_sha"])
    assert Fx.sha(got[1]) == str(g[name + "_dataB_sha"])


def test_offset_depth_rule_matches_the_reference_under_numpy1_and_numpy2_bit_for_bit(se3, golden_dir):
    """`depth -= pose[2,3]*1000` is a float32 operation under the NumPy the reference pins (value-based
    casting) and a float64 one under NumPy 2.  se3tn_set_offset_rule selects; the kernel's tensors are sha256-equal to what the
    reference's own classes produced under NumPy 1.26.4 (preprocess_numpy1.npz, the default rule) and under NumPy 2
    (the *_sha_numpy2 entries / preprocess.npz) -- on whole-millimetre poses (the two agree) and on fractional ones (they do
    not).  The default is the reference's pinned behaviour."""
    from oracle.make_numpy1_golden import OFFSET_CASES
    g = np.load(os.path.join(golden_dir, "preprocess_numpy1.npz"))
    eng = se3.Engine(0, 2)
    mean, std = Fx.mean_std(0)
    eng.set_normalization(mean, std)
    assert eng.get_offset_rule() == "numpy1"
    cases = [(n, s, (t, w)) for n, s, t, w in PRE_CASES] + [(n, s, z) for n, s, z in OFFSET_CASES]
    differ = 0
    for name, seed, extra in cases:
        if isinstance(extra, tuple):
            t, width = extra
            rgb, depth = Fx.synthetic_frame(seed)
            P = Fx.pose(seed, t)
            rgbA, depthA = Fx.synthetic_render(seed + 100, t[2])
            win = se3.crop_window(se3.compute_bbox(P, Fx.K_YCB, width))
        else:
            P = Fx.pose(seed, (0.02, -0.01, extra))
            rgbA, depthA = Fx.synthetic_render(seed + 100, abs(extra))
            rgb, depth = Fx.synthetic_render(seed + 200, abs(extra))
            win = (0, 0, 176, 176)
        r_d, d_d = _frame_to_cuda(rgb, depth)
        ra_d, da_d = _frame_to_cuda(rgbA, depthA)
        z = float(P[2, 3]) * 1000
        got = {}
        for rule in ("numpy1", "numpy2"):
            eng.set_offset_rule(rule)
            out = torch.empty((2, 176, 176, 4), dtype=torch.float32, device="cuda")
            eng.preprocess([dict(rgb=ra_d, depth=da_d, window=(0, 0, 176, 176), z_offset_mm=z, stats=0),
                            dict(rgb=r_d, depth=d_d, window=win, z_offset_mm=z, stats=1)], out)
            torch.cuda.synchronize()
            got[rule] = out.permute(0, 3, 1, 2).contiguous().cpu().numpy()
        assert Fx.sha(got["numpy1"][0]) == str(g[name + "_dataA_sha"]) and Fx.sha(got["numpy1"][1]) == str(g[name + "_dataB_sha"]), name
        assert Fx.sha(got["numpy2"][0]) == str(g[name + "_dataA_sha_numpy2"]) and Fx.sha(got["numpy2"][1]) == str(g[name + "_dataB_sha_numpy2"]), name
        differ += int(not np.array_equal(got["numpy1"], got["numpy2"]))
    assert differ == len(OFFSET_CASES)
    # end to end: what the <= 1-ulp difference does to the network output (far inside the 1e-4 tolerance, reported for DESIGN.md)
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=2)
    m.load_state_dict(sd)
    m.cuda(0)
    P = Fx.pose(21, (0.02, -0.01, 0.8123456789))
    rgbA, depthA = Fx.synthetic_render(121, 0.8123456789)
    rgbB, depthB = Fx.synthetic_render(221, 0.8123456789)
    lg = {}
    for rule in ("numpy1", "numpy2"):
        a, b = O.process_data(rgbA, depthA, P, rgbB, depthB, mean, std, offset_rule=rule)
        m(torch.from_numpy(a)[None].cuda(), torch.from_numpy(b)[None].cuda(), return_feature=False)
        lg[rule] = m.engine.logits(1).cpu().numpy()
    d = float(np.abs(lg["numpy1"] - lg["numpy2"]).max())
    print("OffsetDepth NumPy-1 vs NumPy-2 rounding: max |d logit| = %.2e" % d)
    assert d < 1e-5


def test_pose_update_device_vs_golden(se3, model0, golden_dir):
    """se3tn_infer with poses: compare the device pose composition against processPredict
    applied (oracle) to the device's own (trans, rot)."""
    model, sd = model0
    eng = model.engine
    A, B = Fx.net_inputs(9, 4)
    n = 4
    trans = torch.empty((n, 3), device="cuda"); rot = torch.empty((n, 3), device="cuda")
    poses = np.stack([Fx.pose(70 + i, (0.01 * i, -0.02, 0.6 + 0.1 * i)) for i in range(n)])
    pA = torch.from_numpy(poses.reshape(n, 16)).cuda()
    pB = torch.empty_like(pA)
    for tn, rn in ((0.03, 5 * np.pi / 180), (0.03, 30 * np.pi / 180)):
        eng.set_normalizers(tn, rn)
        eng.infer(A.cuda(), B.cuda(), n, se3.NCHW, trans, rot, pA, pB)
        torch.cuda.synchronize()
        for i in range(n):
            want = O.process_predict(poses[i], trans[i].cpu().numpy(), rot[i].cpu().numpy(), tn, rn)
            got = pB[i].cpu().numpy().reshape(4, 4)
            assert np.abs(got - want).max() < 1e-12, np.abs(got - want).max()
            assert (got[3] == np.array([0, 0, 0, 1.0])).all()
    eng.set_normalizers(0.03, 5 * np.pi / 180)


class _Render:
    """synthetic stand-in for the renderer (same generator as the golden)."""
    def __init__(self):
        self.f = 0

    def render(self, ob2cam, K, window):
        rgbA, depthA = Fx.synthetic_render(130 + self.f, ob2cam[2, 3])
        self.f += 1
        return rgbA, depthA


def test_tracker_on_track_vs_reference_golden(se3, golden_dir):
    """Tracker drop-in: 3 frames with pose feedback, against the goldens produced by the
    composition of the reference's inner functions (predict.py:217-296)."""
    g = np.load(os.path.join(golden_dir, "on_track.npz"))
    sd = O.make_state_dict(0, head_gain=ON_TRACK_HEAD_GAIN)
    mean, std = Fx.mean_std(0)
    trk = se3.Tracker(Fx.DATASET_INFO, mean, std, {"state_dict": sd}, model_path=None, renderer=_Render())
    P = Fx.pose(3)
    for f in range(3):
        rgb, depth = Fx.synthetic_frame(30 + f)
        P = trk.on_track(P, rgb, depth, samples=1)
        assert P.dtype == np.float64 and P.shape == (4, 4)
        assert (trk.last_prediction["bbox"] == g["bbox"][f]).all()          # identical integer bbox track
        assert np.abs(trk.last_prediction["trans"][0] - g["trans"][f]).max() < NET_TOL
        assert np.abs(trk.last_prediction["rot"][0] - g["rot"][f]).max() < NET_TOL
        assert np.abs(P - g["poses"][f + 1]).max() < POSE_TOL, np.abs(P - g["poses"][f + 1]).max()
    assert trk.frame_cnt == 3


def test_f16x3_mode_parity_and_overflow_guard(se3, model0):
    """se3tn_set_precision(F16X3): the 256/512-channel layers run as hi*hi + hi*lo + lo*hi on the f16
    matrix cores.  Same tolerances as the float32 path (measured error stays f32-class), identical
    small-batch results (the latency path stays float32), and the range guard fires on overflow."""
    model, sd = model0
    eng = model.engine
    A, B = Fx.net_inputs(17, 64)
    Ac, Bc = A.cuda(), B.cuda()
    # the float32 side of every comparison below runs the DIRECT kernels: this test is about the two arithmetic modes, the Winograd
    # tiles' own rounding (F(6x6): 6e-6 of a feature map's largest value) is test_winograd_path_vs_direct_and_oracle's subject
    wmin, wtile = eng.get_winograd()
    eng.set_winograd(0)
    o32 = model(Ac, Bc)
    t32, r32, l32, f32 = o32["trans"].clone(), o32["rot"].clone(), eng.logits(64).clone(), o32["feature"].clone()
    eng.set_winograd(wmin, wtile)
    eng.set_precision(se3._lib.PREC_F16X3)
    eng.keep_intermediates(True)      # batch 64 runs the fused Winograd blocks also in this mode: "head" is only stored on request
    try:
        o16 = model(Ac, Bc)
        l16 = eng.logits(64)
        assert not eng.overflow()
        idx = [0, 31, 63]
        ref = O.forward(sd, A[idx], B[idx], intermediates=True)
        _close("f16x3 trans", o16["trans"][idx].cpu(), ref["trans"], 0, NET_TOL)
        _close("f16x3 rot", o16["rot"][idx].cpu(), ref["rot"], 0, NET_TOL)
        _close("f16x3 logits", l16[idx].cpu(), torch.cat([ref["trans_logit"], ref["rot_logit"]], 1), 0, NET_TOL)
        _close("f16x3 feature (decoded split rows)", o16["feature"][idx].cpu(), ref["feature"], ACT_RTOL, 0, 5e-6)
        head = _nchw(eng.debug_buffer("head", 64), 1)[idx]
        _close("f16x3 trans_conv2", head[:, :512], ref["trans_c2"], ACT_RTOL, 0, 5e-6)
        # how far the two arithmetic modes are from each other on the whole batch
        d = float((l16 - l32).abs().max())
        print("max |logit(f16x3) - logit(f32)| over 64 pairs = %.2e" % d)
        assert d < 2e-5
        assert float((o16["feature"] - f32).abs().max()) < 5e-6 * float(f32.abs().max()) + 1e-5
        # small batches take the split-K kernels, also on the f16 matrix cores in this mode
        for nn in (1, 4, 20):
            eng.set_precision(se3._lib.PREC_F32)
            eng.set_winograd(0)
            a = model(Ac[:nn], Bc[:nn])
            a_t, a_f = a["trans"].clone(), a["feature"].clone()
            eng.set_winograd(wmin, wtile)
            eng.set_precision(se3._lib.PREC_F16X3)
            b = model(Ac[:nn], Bc[:nn])
            assert float((a_t - b["trans"]).abs().max()) < 2e-6, nn
            assert float((a_f - b["feature"]).abs().max()) < 5e-6 * float(a_f.abs().max()) + 1e-5, nn
            assert not eng.overflow()
        # range guard: activations beyond the f16 range are reported, not silently wrong
        model(Ac * 3e4, Bc * 3e4, return_feature=False)
        assert eng.overflow()
        assert not eng.overflow()  # reading clears the flag
    finally:
        eng.keep_intermediates(False)
        eng.set_precision(se3._lib.PREC_F32)
        eng.set_winograd(wmin, wtile)


def test_f16x3_split_panels_derived_on_device_equal_host_statement(se3):
    """Blob v7 carries the float32 panels only; se3tn_set_precision(F16X3) derives the split-f16 panels + per-cout scales
    on the device (split_weights_kernel).  Bit-exact against the library's host statement of the same arithmetic (which
    tests/test_host_abi.py pins to an independent numpy restatement from the OIHW weights) -- also after new weights are
    bound while the mode is selected, and for a blob that arrived through bind_blob (the RCCL route)."""
    for seed in (0, 7):
        eng = se3.Engine(0, 4)
        sd = O.make_state_dict(seed)
        if seed == 7:   # awkward rows: a dead cout, a power-of-two maximum, a tiny and a huge row
            w = sd["trans_conv2.conv1.weight"].clone()
            w[3] = 0; w[4, 0, 0, 0] = 4.0; w[5] *= 1e-6; w[6] *= 1e5
            sd["trans_conv2.conv1.weight"] = w
        blob = eng.pack_state_dict(sd)
        want = eng.split_weights_host(blob)
        eng.load_state_dict(sd)
        with pytest.raises(se3._lib.Se3tnError):
            eng.split_weights_device()                      # not derived while the mode was never selected
        eng.set_precision(se3._lib.PREC_F16X3)
        got = eng.split_weights_device()
        assert torch.equal(got, want), int((got != want).sum())
        # new weights while f16x3 is selected: re-derived at upload / bind time
        sd2 = O.make_state_dict(seed + 100)
        want2 = eng.split_weights_host(eng.pack_state_dict(sd2))
        eng.bind_blob(eng.pack_state_dict(sd2).cuda())
        assert torch.equal(eng.split_weights_device(), want2)
        eng.close()


@pytest.mark.parametrize("seed", [101, 102, 103])
def test_f16x3_numerics_under_awkward_scales(se3, seed):
    """The split-f16 path on weights / activations that stress its range handling: per-layer weight
    magnitudes from 1e-3 to 30 (the host scales every cout row by an exact power of two before
    splitting), a dead cout row (all-zero weights), heavy-tailed inputs.  Against the CPU oracle on 2
    pairs and against the float32 kernels on all 32."""
    g = torch.Generator().manual_seed(seed)
    sd = O.make_state_dict(seed, head_gain=1.0)
    facs = {}
    for k in list(sd.keys()):
        if k.endswith("weight") and sd[k].dim() == 4:
            f = float(10 ** (torch.rand(1, generator=g) * 4.5 - 3.0))   # 1e-3 .. 30
            sd[k] = sd[k] * f
            facs[k] = f
            # keep activations O(1): the BN that follows absorbs the scale through its running statistics
            bn = k.replace(".0.weight", ".1.weight").replace("conv1.weight", "bn1.weight").replace("conv2.weight", "bn2.weight")
            sd[k.replace("weight", "bias")] = sd[k.replace("weight", "bias")] * f
            rm = bn.replace("weight", "running_mean"); rv = bn.replace("weight", "running_var")
            sd[rm] = sd[rm] * f; sd[rv] = sd[rv] * f * f
    sd["trans_conv2.conv1.weight"][7] = 0.0          # a dead output channel
    sd["trans_out.0.weight"] *= 0.02; sd["rot_out.0.weight"] *= 0.02
    n = 32
    A = torch.randn((n, 4, 176, 176), generator=g) * (1.0 + 20.0 * (torch.rand((n, 4, 176, 176), generator=g) > 0.999))
    B = torch.randn((n, 4, 176, 176), generator=g) * 3.0
    m = se3.Se3TrackNet(176, max_batch=n)
    m.load_state_dict(sd); m.cuda(0)
    eng = m.engine
    m(A.cuda(), B.cuda(), return_feature=False)
    l32 = eng.logits(n).clone()              # float32, default algorithms: n = 32 -> Winograd blocks (the default tile)
    wmin, wtile = eng.get_winograd()
    eng.set_winograd(0)
    m(A.cuda(), B.cuda(), return_feature=False)
    l32d = eng.logits(n).clone()             # float32, direct kernels only
    eng.set_winograd(wmin, wtile)
    assert not torch.equal(l32, l32d)
    eng.set_precision(se3._lib.PREC_F16X3)
    m(A.cuda(), B.cuda(), return_feature=False)
    l16 = eng.logits(n).clone()
    assert not eng.overflow()
    ref = O.forward(sd, A[:2], B[:2])
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
    scale = float(want.abs().max()) + 1.0
    e32 = float((l32[:2].cpu() - want).abs().max()) / scale
    e32d = float((l32d[:2].cpu() - want).abs().max()) / scale
    e16 = float((l16[:2].cpu() - want).abs().max()) / scale
    d = float((l16 - l32d).abs().max()) / scale
    dw = float((l32 - l32d).abs().max()) / scale
    print("seed %d: rel err f32 direct %.2e  f32 Winograd %.2e  f16x3 %.2e  |f16x3 - direct| %.2e  |Winograd - direct| %.2e "
          "(logit scale %.2f)" % (seed, e32d, e32, e16, d, dw, scale))
    assert e32 < 2e-5 and e32d < 2e-5 and e16 < 2e-5 and d < 2e-5 and dw < 2e-5
