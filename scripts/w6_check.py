"""Dev check of the Winograd F(6x6,3x3) path (GPU box): logits vs the direct kernels, F(4x4) and the oracle at n = 3 and n = 16."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O
sd = O.make_state_dict(0)
for n in (3, 16):
    m = se3.Se3TrackNet(176, max_batch=16); m.load_state_dict(sd); m.cuda(0)
    A, B = Fx.net_inputs(11, n); Ac, Bc = A.cuda(), B.cuda()
    res = {}
    for name, (mb, tile) in (("direct", (0, 0)), ("F(4x4)", (1, 4)), ("F(6x6)", (1, 6))):
        m.engine.set_winograd(mb, tile)
        o = m(Ac, Bc)
        res[name] = (m.engine.logits(n).cpu().clone(), o["feature"].cpu().clone())
    ref = O.forward(sd, A[:3], B[:3])
    want = torch.cat([ref["trans_logit"], ref["rot_logit"]], 1)
    for name, (l, f) in res.items():
        print("n=%d %-7s |logit - oracle| %.3e   |logit - direct| %.3e   |feature - direct| / max %.3e" % (
            n, name, float((l[:3] - want).abs().max()), float((l - res["direct"][0]).abs().max()),
            float((f - res["direct"][1]).abs().max() / res["direct"][1].abs().max())))
