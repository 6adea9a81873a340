"""Dev check of the fused trunk Winograd path (GPU box): parity vs the direct kernels + the oracle, and per-launch times."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O
sd = O.make_state_dict(0)
for n in (64, 8, 20):
    m = se3.Se3TrackNet(176, max_batch=64); m.load_state_dict(sd); m.cuda(0)
    A, B = Fx.net_inputs(11, n); Ac, Bc = A.cuda(), B.cuda()
    m.engine.set_trunk_winograd(0)
    m(Ac, Bc); l0 = m.engine.logits(n).clone(); q0 = m.engine.debug_buffer("q64", n).clone(); t0 = m.engine.debug_buffer("t64", n).clone()
    m.engine.set_trunk_winograd(1, 0)
    m(Ac, Bc); l1 = m.engine.logits(n).clone(); q1 = m.engine.debug_buffer("q64", n).clone(); t1 = m.engine.debug_buffer("t64", n).clone()
    ref = O.forward(sd, A[:2], B[:2])
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
    print("n=%d  |q64 fused - direct| max %.3e (max |q64| %.2f)   t64 %.3e   |logit fused - direct| %.3e   vs oracle: direct %.3e fused %.3e" % (
        n, float((q1 - q0).abs().max()), float(q0.abs().max()), float((t1 - t0).abs().max()), float((l1 - l0).abs().max()),
        float((l0[:2].cpu() - want).abs().max()), float((l1[:2].cpu() - want).abs().max())))
