"""The one REAL image pair the reference repository ships (media/0000000rgbA.png, 0000000rgbB.png: its README's example crops) through
the reference's own TrackDataset.processData + Se3TrackNet (tests/golden/media_pair.npz, oracle/make_media_golden.py), against the
oracle (CPU) and the HIP path (GPU): pre-processing bit for bit, (trans, rot) within 1e-4, pose within 1e-5."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as Fx
from oracle import se3_oracle as O


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "media_pair.npz"))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_real_images_are_what_the_reference_repository_ships(golden):
    a, b = golden["rgbA"], golden["rgbB"]
    assert a.shape == b.shape == (176, 176, 3) and a.dtype == np.uint8
    assert 0.2 < (a.sum(-1) > 0).mean() < 0.3 and (b.sum(-1) > 0).mean() > 0.6      # a rendered silhouette / an observed crop
    path = "/root/reference/media/0000000rgbA.png"
    if os.path.isfile(path):
        from PIL import Image
        assert np.array_equal(np.array(Image.open(path)), a)
        assert np.array_equal(np.array(Image.open(path.replace("rgbA", "rgbB"))), b)


def test_oracle_on_the_real_pair_equals_the_reference(golden):
    mean, std = Fx.mean_std(0)
    a, b = O.process_data(golden["rgbA"], golden["depthA"], golden["pose"], golden["rgbB"], golden["depthB"], mean, std, offset_rule="numpy2")
    assert _sha(a) == str(golden["dataA_sha"]) and _sha(b) == str(golden["dataB_sha"])
    sd = O.make_state_dict(0, head_gain=float(golden["head_gain"]))
    out = O.forward(sd, torch.from_numpy(a)[None], torch.from_numpy(b)[None])
    assert np.abs(out["trans"][0].numpy() - golden["trans"]).max() < 2e-6 and np.abs(out["rot"][0].numpy() - golden["rot"]).max() < 2e-6
    assert 0.03 < np.abs(golden["trans"]).min() and np.abs(np.r_[golden["trans"], golden["rot"]]).max() < 0.5       # not saturated
    P = O.process_predict(golden["pose"], golden["trans"], golden["rot"], 0.03, 5 * np.pi / 180)
    assert np.abs(P - golden["poseB"]).max() < 1e-12


@pytest.mark.gpu
def test_hip_path_on_the_real_pair(golden):
    import se3tracknet_amd as se3
    mean, std = Fx.mean_std(0)
    sd = O.make_state_dict(0, head_gain=float(golden["head_gain"]))
    eng = se3.Engine(0, 1)
    eng.load_state_dict(sd)
    eng.set_normalization(mean, std)
    eng.set_normalizers(0.03, 5 * np.pi / 180)
    eng.set_offset_rule("numpy2")                       # the golden: the reference under this image's NumPy 2
    dev = "cuda:0"
    z_mm = float(golden["pose"][2, 3]) * 1000

    def crop(rgb, depth, stats):
        return dict(rgb=torch.from_numpy(np.ascontiguousarray(rgb)).to(dev),
                    depth=torch.from_numpy(np.ascontiguousarray(depth).view(np.int16)).to(dev), window=(0, 0, 176, 176), z_offset_mm=z_mm,
                    stats=stats)
    outA = torch.empty((1, 176, 176, 4), device=dev)
    outB = torch.empty((1, 176, 176, 4), device=dev)
    eng.preprocess([crop(golden["rgbA"], golden["depthA"], 0)], outA)
    eng.preprocess([crop(golden["rgbB"], golden["depthB"], 1)], outB)
    a = outA[0].permute(2, 0, 1).contiguous().cpu().numpy()
    b = outB[0].permute(2, 0, 1).contiguous().cpu().numpy()
    assert _sha(a) == str(golden["dataA_sha"]) and _sha(b) == str(golden["dataB_sha"])          # pre-processing: every bit
    trans = torch.empty((1, 3), device=dev); rot = torch.empty((1, 3), device=dev)
    poseA = torch.from_numpy(golden["pose"].reshape(1, 16)).to(dev)
    poseB = torch.empty((1, 16), dtype=torch.float64, device=dev)
    eng.infer(outA, outB, 1, se3.NHWC, trans, rot, poseA, poseB)
    torch.cuda.synchronize()
    d = max(float(np.abs(trans.cpu().numpy()[0] - golden["trans"]).max()), float(np.abs(rot.cpu().numpy()[0] - golden["rot"]).max()))
    dp = float(np.abs(poseB.cpu().numpy().reshape(4, 4) - golden["poseB"]).max())
    print("real image pair: max |d(trans, rot)| %.2e, max |d pose| %.2e vs the reference modules" % (d, dp))
    assert d < 1e-4 and dp < 1e-5
