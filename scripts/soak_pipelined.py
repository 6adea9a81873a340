#!/usr/bin/env python3
"""Soak test of the throughput mode (run on the GPU box): N batches alternating over the two lanes of a
PipelinedEngine, queued without host synchronisation, EVERY batch's (trans, rot, pose) compared on the device with the
single-stream result.  One line per arithmetic mode.   python scripts/soak_pipelined.py [steps] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

import se3tracknet_amd as se3
from oracle import fixtures as Fx
from oracle import se3_oracle as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
sd = O.make_state_dict(0)
A, B = Fx.net_inputs(5, n)
Ac, Bc = A.cuda(), B.cuda()
poseA = torch.eye(4, dtype=torch.float64, device="cuda").repeat(n, 1, 1).contiguous()
for mode in ("f32", "f16x3", "direct"):
    pe = se3.PipelinedEngine(0, n, depth=2)
    pe.load_state_dict(sd)
    pe.set_normalizers(0.03, 5 * np.pi / 180)
    if mode == "f16x3":
        pe.set_precision(se3._lib.PREC_F16X3)
    elif mode == "direct":
        pe.set_winograd(0)
        pe.set_trunk_winograd(0)
    ref = [torch.empty((n, 3), device="cuda"), torch.empty((n, 3), device="cuda"), torch.empty_like(poseA)]
    pe.engines[0].infer(Ac, Bc, n, se3.NCHW, ref[0], ref[1], poseA, ref[2])
    torch.cuda.synchronize()
    outs = [[torch.empty((n, 3), device="cuda"), torch.empty((n, 3), device="cuda"), torch.empty_like(poseA)] for _ in range(2)]
    bad = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
    t0 = time.perf_counter()
    for i in range(steps):
        eng, stream = pe.next_lane()
        k = pe.engines.index(eng)
        with torch.cuda.stream(stream):
            eng.infer(Ac, Bc, n, se3.NCHW, outs[k][0], outs[k][1], poseA, outs[k][2])
            differs = (outs[k][0] != ref[0]).any() | (outs[k][1] != ref[1]).any() | (outs[k][2] != ref[2]).any()
            bad[k] += differs.to(torch.int64)
    pe.synchronize()
    dt = time.perf_counter() - t0
    nbad = int(bad[0].item() + bad[1].item())
    print("%-6s batch %d: %d batches over 2 lanes, %d differ from the single-stream result (%.1f k pairs/s incl. the checks)" % (
        mode, n, steps, nbad, steps * n / dt / 1e3))
