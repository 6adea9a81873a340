"""GPU: every batch size around the algorithm / tile-shape switch points against the CPU oracle.
Switch points of se3tn_infer (n pairs): split-K latency kernels below ~200 big-tile workgroups per layer, the Winograd
blocks (default: F(4x4) fused blocks for 6 <= n < 14, F(6x6) conv by conv from n = 14; F(4x4) at every n as the second
parametrisation) from n >= 6, the fused Winograd F(2x2) trunk kernel per launch from n = 18 (grouped A2|B2 launches) / n = 34 (B3) wherever
the rounds of workgroups are >= 55 % full (of the sizes below: 48..52, 63, 64 and 72 run it in all four trunk launches, 25..31 and
66..69 in the two grouped ones, 33 and everything up to 17 in none), 256 x 128 stride-2 tiles while they fill 200..256 CUs (n = 50..68 for the heads, 26..67 for convAB1), ragged last
tiles at every n that is not a multiple of the tile size."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as Fx
from oracle import se3_oracle as O

SIZES = [1, 2, 3, 5, 6, 7, 9, 13, 17, 25, 26, 27, 31, 33, 48, 49, 50, 52, 63, 64, 66, 67, 68, 69, 72]


@pytest.mark.parametrize("tile", [0, 4], ids=["default tile", "F(4x4) fused blocks"])
def test_every_switch_point_vs_oracle(tile):
    import se3tracknet_amd as se3
    sd = O.make_state_dict(3)
    m = se3.Se3TrackNet(176, max_batch=72)
    m.load_state_dict(sd)
    m.cuda(0)
    if tile:
        m.engine.set_winograd(m.engine.get_winograd()[0], tile)
    A, B = Fx.net_inputs(77, 72)
    Ac, Bc = A.cuda(), B.cuda()
    idx = [0, 35, 71]
    ref = O.forward(sd, A[idx], B[idx])
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
    worst = 0.0
    for n in SIZES:
        # pairs 0 and n-1 of this call are pairs (0 | 35 | 71) of the full set, depending on how the slice is taken
        for lo in (0, 72 - n):
            out = m(Ac[lo:lo + n], Bc[lo:lo + n], return_feature=False)
            lg = m.engine.logits(n).cpu()
            for k, gi in enumerate(idx):
                if lo <= gi < lo + n:
                    e = float((lg[gi - lo] - want[k]).abs().max())
                    worst = max(worst, e)
                    assert e < 1e-4, (n, lo, gi, e)
            assert torch.isfinite(out["trans"]).all()
    print("max |d logit| over %d batch sizes: %.2e" % (len(SIZES), worst))


def test_f16x3_mode_every_switch_point_vs_oracle():
    """The same sweep in SE3TN_PREC_F16X3: split-K f16 kernels below the big-tile threshold, direct f16x3 kernels, and from
    n >= 6 the 512-channel head block as fused Winograd F(4x4) passes on split-f16 operands (ragged last GEMM tiles at every n
    that is not a multiple of 32 / 3 tiles); the range guard must stay silent and a mode switch back must restore float32."""
    import se3tracknet_amd as se3
    sd = O.make_state_dict(4)
    m = se3.Se3TrackNet(176, max_batch=72)
    m.load_state_dict(sd)
    m.cuda(0)
    A, B = Fx.net_inputs(78, 72)
    Ac, Bc = A.cuda(), B.cuda()
    idx = [0, 35, 71]
    ref = O.forward(sd, A[idx], B[idx])
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
    out32 = m(Ac[:9], Bc[:9], return_feature=False)["trans"].clone()
    m.engine.set_precision(se3._lib.PREC_F16X3)
    worst = 0.0
    try:
        for n in [1, 5, 6, 7, 9, 17, 31, 33, 49, 64, 67, 72]:
            for lo in (0, 72 - n):
                out = m(Ac[lo:lo + n], Bc[lo:lo + n], return_feature=False)
                lg = m.engine.logits(n).cpu()
                for k, gi in enumerate(idx):
                    if lo <= gi < lo + n:
                        e = float((lg[gi - lo] - want[k]).abs().max())
                        worst = max(worst, e)
                        assert e < 1e-4, (n, lo, gi, e)
                assert torch.isfinite(out["trans"]).all() and not m.engine.overflow()
    finally:
        m.engine.set_precision(se3._lib.PREC_F32)
    print("f16x3: max |d logit| over the sweep: %.2e" % worst)
    assert worst < 2e-5
    assert torch.equal(m(Ac[:9], Bc[:9], return_feature=False)["trans"], out32)
