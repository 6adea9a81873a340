"""GPU: 300-frame closed-loop track (stand-in for BASELINE configs[2]) through the drop-in
Tracker with the HIP rasteriser, every frame checked against the CPU oracle fed the same rendered image:
identical integer bbox track, (trans, rot) within 1e-4, pose within 1e-5."""
import pytest

pytestmark = pytest.mark.gpu

from oracle import closed_loop


@pytest.fixture(scope="module")
def se3():
    import se3tracknet_amd
    return se3tracknet_amd


def test_closed_loop_300_frames_per_frame_parity(se3):
    r = closed_loop.run(se3, frames=300, check=True, timing=False)
    print(r)
    assert r["frames_checked"] == 300
    assert r["bbox_mismatches"] == 0, r
    assert r["max_abs_trans_rot"] <= 1e-4 and r["max_abs_pose"] <= 1e-5, r
    assert r["max_drift_m"] > 0.002, "the pose never moved: the feedback loop is not exercised"
    assert r["reinits_checked_pass"] == 0
    assert r["ok"]


def test_closed_loop_f16x3_mode(se3):
    r = closed_loop.run(se3, frames=60, check=True, timing=False, precision=se3._lib.PREC_F16X3)
    assert r["bbox_mismatches"] == 0 and r["max_abs_trans_rot"] <= 1e-4 and r["max_abs_pose"] <= 1e-5, r
