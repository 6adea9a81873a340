"""CPU: the LDS slab of conv3x3_slab_kernel (csrc/conv3x3_mfma.hip) must hold, for every tile of 256
consecutive output pixels, the padded-flat input run [lo, hi) the 9 taps touch.  This brute-forces
the worst case per layer geometry against the SLABPX template arguments in launch_conv3x3."""
import re
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def padded_index(m, H, W):
    n, rem = divmod(m, H * W)
    h, w = divmod(rem, W)
    return (n * (H + 2) + h + 1) * (W + 2) + w + 1


def slab_pixels(H, W, batch, BM=256):
    M = batch * H * W
    worst = 0
    for m0 in range(0, M, BM):
        mlast = min(m0 + BM, M) - 1
        lo = padded_index(m0, H, W) - (W + 3)
        hi = padded_index(mlast, H, W) + (W + 3) + 1
        assert lo >= 0 and hi <= batch * (H + 2) * (W + 2)
        # all taps of all tile pixels stay inside [lo, hi)
        worst = max(worst, -(-(hi - lo) // 8) * 8)
    return worst


@pytest.mark.parametrize("W,expected_max", [(44, 464), (22, 384), (11, 424)])
def test_slab_template_sizes_cover_worst_case(W, expected_max):
    src = open(os.path.join(ROOT, "iros20-6d-pose-tracking_amd", "csrc", "conv3x3_mfma.hip")).read()
    sizes = set(int(x) for x in re.findall(r"launch_slab<\d+, \d, \d, \d, \d, (\d+), \d>", src))
    assert expected_max in sizes
    for batch in (1, 2, 3, 5, 64, 67):
        assert slab_pixels(W, W, batch) <= expected_max, (W, batch, slab_pixels(W, W, batch))
    # pieces are handed out as wid + 8 t, t < 9  ->  at most 72 pieces of 8 pixels
    assert expected_max // 8 <= 72


def test_taps_are_shifted_windows_of_the_padded_run():
    """The property the slab kernel rests on: with the zero border stored in memory, tap (r,s) of
    interior pixel m is the padded-flat pixel padded_index(m) + (r-1)*(W+2) + (s-1)."""
    H = W = 5
    rng = np.random.default_rng(0)
    x = np.zeros((2, H + 2, W + 2)); x[:, 1:-1, 1:-1] = rng.normal(size=(2, H, W))
    flat = x.reshape(-1)
    for m in range(2 * H * W):
        n, rem = divmod(m, H * W); h, w = divmod(rem, W)
        for r in range(3):
            for s in range(3):
                hi, wi = h + r - 1, w + s - 1
                want = x[n, hi + 1, wi + 1]  # zero outside the image
                assert flat[padded_index(m, H, W) + (r - 1) * (W + 2) + (s - 1)] == want
