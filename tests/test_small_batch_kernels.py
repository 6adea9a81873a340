"""The batch 1-5 kernel family (stem_pool_small, conv64_small, conv_slices_small: the regime Tracker.on_track runs in, predict.py:416)
against the CPU oracle and against the general kernels (split-K / batch-64 stem + pool) on the same inputs.
  * every n in 1..5: pre-tanh logits within the north-star tolerance of the oracle, within 2e-5 of the general kernels;
  * bitwise reproducible run to run;
  * which kernels run: the profile names them;
  * every n in 1..5: a pair has the same bits alone and as one of n (every kernel of the family works image by image with a
    layer-fixed summation order; round 6: stem_pool_small serves the whole family);
  * every intermediate map of one pair against the oracle (the kernels' outputs, not only the regression)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fixtures as Fx
from oracle import se3_oracle as O

NET_TOL = 1e-4


@pytest.fixture(scope="module")
def se3():
    import se3tracknet_amd
    return se3tracknet_amd


def _logits(m, A, B, n):
    m(A, B, return_feature=False)
    torch.cuda.synchronize()
    return m.engine.logits(n).cpu().numpy()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_small_batch_family_vs_oracle_and_general_kernels(se3, n):
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=n)
    m.load_state_dict(sd)
    m.cuda(0)
    eng = m.engine
    A, B = Fx.net_inputs(7, n)
    Ac, Bc = A.cuda(), B.cuda()
    assert eng.get_small_kernels()
    small = _logits(m, Ac, Bc, n)
    assert np.array_equal(small, _logits(m, Ac, Bc, n)), "not reproducible run to run"
    eng.profile_enable(1)
    m(Ac, Bc, return_feature=False)
    torch.cuda.synchronize()
    names = [nm for nm, _ in eng.profile_launches(0)]
    eng.profile_enable(0)
    assert any("small tiles" in nm for nm in names), names                      # stem + pool in one launch for the whole family (1-5 pairs)
    assert "maxpool3x3s2" not in names, names
    eng.set_small_kernels(False)
    general = _logits(m, Ac, Bc, n)
    eng.set_small_kernels(True)
    ref = O.forward(sd, A, B)
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1).numpy()
    d_gen, d_ref = float(np.abs(small - general).max()), float(np.abs(small - want).max())
    print("n = %d: max |d logits| vs the general kernels %.2e, vs the oracle %.2e" % (n, d_gen, d_ref))
    assert d_gen < 2e-5 and d_ref < NET_TOL
    # a pair alone through the same family
    m1 = se3.Se3TrackNet(176, max_batch=1)
    m1.load_state_dict(sd)
    m1.cuda(0)
    alone = np.concatenate([_logits(m1, Ac[i:i + 1], Bc[i:i + 1], 1) for i in range(n)])
    # ONE family with layer-fixed summation orders for every n <= 5 (round 6: the stem + pool kernel no longer changes at 3 pairs)
    assert np.array_equal(alone, small), "1-5 pairs: a pair's bits must not depend on the batch it travels in"


def test_small_batch_family_intermediates_vs_oracle(se3):
    """what the kernels of the family WRITE: the pooled stem map of branch A (stem_pool_small: never stores the stem map itself) and
    the 256-channel feature after convAB2 (conv64_small x 4, conv_slices_small x 3 and their reductions behind it)"""
    sd = O.make_state_dict(0)
    m = se3.Se3TrackNet(176, max_batch=2)
    m.load_state_dict(sd)
    m.cuda(0)
    eng = m.engine
    A, B = Fx.net_inputs(3, 2)
    m(A.cuda(), B.cuda(), return_feature=False)
    torch.cuda.synchronize()
    ref = O.forward(sd, A, B, intermediates=True)
    pool = eng.debug_buffer("pool", 2)[:, 1:-1, 1:-1, :64].permute(0, 3, 1, 2).cpu()
    ab = eng.debug_buffer("ab", 2)[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).cpu()
    for name, got, want in (("poolA", pool, ref["poolA"]), ("feature", ab, ref["feature"])):
        scale = max(float(want.abs().max()), 1.0)
        err = float((got - want).abs().max())
        print("%s: max abs err %.2e (scale %.2f)" % (name, err, scale))
        assert got.shape == want.shape and err <= 2e-5 * scale, name
