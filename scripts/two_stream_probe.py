#!/usr/bin/env python3
"""Does pipelining successive batch-64 steps over two HIP streams (two contexts, alternate steps) hide the HBM-bound
passes of one step under the MFMA-bound kernels of the other?  Prints pairs/s for 1, 2 and 3 streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import se3tracknet_amd as se3
from oracle import se3_oracle as O

nb, H, W = 64, 480, 640
dev = "cuda:0"
sd = O.make_state_dict(0)
mean = np.array([110., 105., 100., 1000., 112., 104., 99., 1010.]); std = np.array([60., 58., 61., 300., 59., 60., 62., 310.])
g = torch.Generator(device=dev).manual_seed(1234)
frames_rgb = torch.randint(0, 256, (nb, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
frames_d = torch.randint(300, 1500, (nb, H, W), generator=g, device=dev, dtype=torch.int16)
rend_rgb = torch.randint(0, 256, (nb, 176, 176, 3), generator=g, device=dev, dtype=torch.uint8)
rend_d = torch.randint(600, 1000, (nb, 176, 176), generator=g, device=dev, dtype=torch.int16)
rng = np.random.default_rng(7)
poses = np.tile(np.eye(4), (nb, 1, 1)); poses[:, 0, 3] = rng.uniform(-0.15, 0.15, nb); poses[:, 1, 3] = rng.uniform(-0.1, 0.1, nb)
poses[:, 2, 3] = rng.uniform(0.6, 1.0, nb)
K = np.array([[1066.778, 0, 312.9869], [0, 1067.487, 241.3109], [0, 0, 1]])
wB = np.array([se3.crop_window(se3.compute_bbox(poses[i], K, 250.0)) for i in range(nb)])
wA = np.tile(np.array([0, 0, 176, 176]), (nb, 1)); z = poses[:, 2, 3] * 1000.0
cA = se3.pack_crops(rend_rgb, rend_d, wA, z, 0); cB = se3.pack_crops(frames_rgb, frames_d, wB, z, 1)
poseA = torch.from_numpy(poses.reshape(nb, 16)).to(dev)

for ns in (1, 2, 3, 1, 2):
    engs, streams, outs = [], [], []
    for s in range(ns):
        e = se3.Engine(0, nb); e.load_state_dict(sd); e.set_normalization(mean, std); e.set_normalizers(0.03, 5 * np.pi / 180)
        engs.append(e); streams.append(torch.cuda.Stream(device=dev))
        outs.append((torch.empty((nb, 3), device=dev), torch.empty((nb, 3), device=dev), torch.empty_like(poseA)))
    def step(i):
        k = i % ns
        with torch.cuda.stream(streams[k]):
            e = engs[k]
            e.preprocess(cA, e.input_buffer_ptr(0)); e.preprocess(cB, e.input_buffer_ptr(1))
            e.infer(e.input_buffer_ptr(0), e.input_buffer_ptr(1), nb, se3.NHWC, outs[k][0], outs[k][1], poseA, outs[k][2])
    for i in range(12): step(i)
    torch.cuda.synchronize()
    steps = 600
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d stream(s): %8.1f pairs/s  %.4f ms/step  (poses equal across contexts: %s)" % (
        ns, nb * steps / dt, dt / steps * 1e3, all(torch.equal(outs[0][2], o[2]) for o in outs)))
    del engs
