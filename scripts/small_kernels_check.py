"""The batch 1-5 kernel family (conv64_small, conv_slices_small, stem_pool_small) against the split-K / batch-64 kernels and the
oracle: max |d logits|, bitwise reproducibility, and that a pair's bits do not depend on the batch it travels in."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import se3tracknet_amd as se3
from oracle import fixtures as Fx, se3_oracle as O

sd = O.make_state_dict(0)
ok = True
for n in (1, 2, 3, 5):
    m = se3.Se3TrackNet(176, max_batch=n); m.load_state_dict(sd); m.cuda(0)
    A, B = Fx.net_inputs(7, n); Ac, Bc = A.cuda(), B.cuda()
    eng = m.engine
    outs = {}
    for small in (True, False):
        eng.set_small_kernels(small)
        m(Ac, Bc, return_feature=False); torch.cuda.synchronize()
        outs[small] = eng.logits(n).cpu().numpy()
        m(Ac, Bc, return_feature=False); torch.cuda.synchronize()
        assert np.array_equal(outs[small], eng.logits(n).cpu().numpy()), "not reproducible"
    d = float(np.abs(outs[True] - outs[False]).max())
    eng.set_small_kernels(True)
    alone = []
    m1 = se3.Se3TrackNet(176, max_batch=1); m1.load_state_dict(sd); m1.cuda(0)
    for i in range(n):
        m1(Ac[i:i + 1], Bc[i:i + 1], return_feature=False); torch.cuda.synchronize()
        alone.append(m1.engine.logits(1).cpu().numpy())
    alone = np.concatenate(alone)
    same = bool(np.array_equal(alone, outs[True]))
    print("n = %d: max |d logits| small vs split-K kernels %.3e; pair alone == pair in the batch: %s (max diff %.1e)" % (
        n, d, same, float(np.abs(alone - outs[True]).max())))
    ok &= d < 2e-5
print("OK" if ok else "FAILED")
