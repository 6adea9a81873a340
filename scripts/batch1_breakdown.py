import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O
for n in (1, 2, 4):
    m = se3.Se3TrackNet(176, max_batch=n); m.load_state_dict(O.make_state_dict(0)); m.cuda(0)
    A, B = Fx.net_inputs(1, n); Ac, Bc = A.cuda(), B.cuda()
    for _ in range(5): m(Ac, Bc, return_feature=False)
    eng = m.engine
    eng.profile_enable(8)
    for _ in range(8): m(Ac, Bc, return_feature=False)
    torch.cuda.synchronize()
    acc = {}
    for s in range(8):
        for nm, ms in eng.profile_launches(s):
            acc.setdefault(nm, []).append(ms)
    tot = 0
    print("n =", n)
    for nm, v in acc.items():
        print("   %-46s %7.1f us" % (nm, 1e3 * float(np.median(v)))); tot += float(np.median(v))
    print("   total %.1f us" % (tot * 1e3))
    eng.profile_enable(0)
