"""A REAL image pair through the reference (TEST INFRASTRUCTURE ONLY; build container):

    python -m oracle.make_media_golden        ->  tests/golden/media_pair.npz

media/0000000rgbA.png / 0000000rgbB.png of the reference repository are the example pair of its README: a rendered crop A and an
observed crop B, 176 x 176 RGB (the only real images the repository ships; every other fixture here is synthetic).  No depth comes
with them: depth A is a seeded smooth surface under the rendered silhouette, depth B a tilted plane with bumps, dropouts and far
values (so every OffsetDepth branch is hit).  The pair goes through the reference's OWN TrackDataset.processData
(datasets.py:115-156: OffsetDepth, NormalizeChannels, ToTensor) and Se3TrackNet (random-init, seeded) imported unmodified; stored:
the inputs (the two images as arrays + the depths: the GPU box has no /root/reference), sha256 + sub-sample of dataA / dataB, and
the network outputs.  NumPy generation of this interpreter (2.x) -> offset rule 'numpy2'."""
import hashlib
import os

import numpy as np
import torch
from PIL import Image

from . import fixtures as Fx
from . import ref_shims
from . import se3_oracle as O
from .make_golden import ref_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MEDIA = os.path.join(ref_shims.REFERENCE_ROOT, "media")
HEAD_GAIN = 0.0004
POSE_Z = 0.8123456789


def synth_depths(rgbA, rgbB, seed=77):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:176, 0:176].astype(np.float64)
    r2 = ((xx - 88) / 60.0) ** 2 + ((yy - 88) / 60.0) ** 2
    dA = np.where(rgbA.sum(-1) > 0, 1000 * POSE_Z - 40 * np.sqrt(np.clip(1 - r2, 0, 1)), 0).astype(np.uint16)
    dB = 1000 * POSE_Z - 30 + 0.4 * (xx - 88) + 0.2 * (yy - 88) + 12 * np.sin(xx / 9.0) * np.cos(yy / 11.0) + rng.normal(0, 1.5, (176, 176))
    dB = np.clip(dB, 0, 65535)
    dB[rng.uniform(size=dB.shape) < 0.03] = 0            # dropouts
    dB[20:30, 140:170] = 2600                              # beyond the 2000 mm validity bound
    dB[150:160, 10:40] = 60                                # below the 100 mm bound
    return dA, dB.astype(np.uint16)


def main():
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    ref = ref_shims.load()
    U, DA, DS = ref.Utils, ref.data_augmentation, ref.datasets
    rgbA = np.array(Image.open(os.path.join(MEDIA, "0000000rgbA.png")))
    rgbB = np.array(Image.open(os.path.join(MEDIA, "0000000rgbB.png")))
    assert rgbA.shape == rgbB.shape == (176, 176, 3) and rgbA.dtype == np.uint8
    depthA, depthB = synth_depths(rgbA, rgbB)
    mean, std = Fx.mean_std(0)
    sd = O.make_state_dict(0, head_gain=HEAD_GAIN)
    model = ref_model(ref, sd)
    post = U.Compose([DA.OffsetDepth(), DA.NormalizeChannels(mean, std), DA.ToTensor()])
    ds = DS.TrackDataset('', 'eval', mean, std, None, None, post, Fx.DATASET_INFO, trans_normalizer=0.03, rot_normalizer=5 * np.pi / 180)
    P = Fx.pose(9, (0.03, -0.02, POSE_Z))
    sample = ds.processData(rgbA, depthA, P, rgbB, depthB, np.eye(4))[0]
    a, b = sample[0].numpy().astype(np.float32), sample[1].numpy().astype(np.float32)
    with torch.no_grad():
        pred = model(sample[0].unsqueeze(0).float(), sample[1].unsqueeze(0).float())
    t, r = pred["trans"][0].numpy(), pred["rot"][0].numpy()
    poseB = ds.processPredict(P, (t, r))
    sha = lambda x: hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()   # noqa: E731
    out = dict(rgbA=rgbA, rgbB=rgbB, depthA=depthA, depthB=depthB, pose=P, dataA_sha=np.array(sha(a)), dataB_sha=np.array(sha(b)),
               dataA_sub=a[:, ::7, ::7].copy(), dataB_sub=b[:, ::7, ::7].copy(), trans=t, rot=r, poseB=poseB,
               numpy_version=np.array(np.__version__), head_gain=np.float64(HEAD_GAIN))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "media_pair.npz"), **out)
    print("media_pair.npz: trans %s rot %s; covered A %.2f, valid B %.2f" % (t, r, (depthA > 0).mean(), ((depthB > 100) & (depthB < 2000)).mean()))


if __name__ == "__main__":
    main()
