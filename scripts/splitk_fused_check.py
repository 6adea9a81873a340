"""Run on the GPU box: the fused split-K reduction against the separate conv_reduce_kernel launch (SE3TN_SPLITK_FUSED=0), bitwise,
at the small batch sizes that take the split-K path, both arithmetic modes; then 2000 repeated forwards (counters re-arm)."""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
import numpy as np

CHILD = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O
out = {}
for mode in ("f32", "f16x3"):
    for n in (1, 2, 3, 4, 5):
        m = se3.Se3TrackNet(176, max_batch=n); m.load_state_dict(O.make_state_dict(0)); m.cuda(0)
        if mode == "f16x3": m.engine.set_precision(se3._lib.PREC_F16X3)
        A, B = Fx.net_inputs(7, n)
        r = m(A.cuda(), B.cuda())
        lg = m.engine.logits(n).cpu().numpy().copy()
        for _ in range(400 if n == 1 else 50):
            m(A.cuda(), B.cuda(), return_feature=False)
        torch.cuda.synchronize()
        lg2 = m.engine.logits(n).cpu().numpy()
        assert np.array_equal(lg, lg2), (mode, n, "not reproducible run to run")
        out["%s_%d" % (mode, n)] = lg
np.savez(sys.argv[1], **out)
'''
res = {}
for fused in ("1", "0"):
    path = "/tmp/splitk_%s.npz" % fused
    env = dict(os.environ, SE3TN_SPLITK_FUSED=fused)
    subprocess.run(["timeout", "300", sys.executable, "-c", CHILD, path], check=True, env=env)
    res[fused] = np.load(path)
bad = [k for k in res["1"].files if not np.array_equal(res["1"][k], res["0"][k])]
print("fused vs separate reduce: %d configurations, %d differ %s" % (len(res["1"].files), len(bad), bad))
sys.exit(1 if bad else 0)
